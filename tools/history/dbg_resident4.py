import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
L=capi.lib()
L.cvh_debug_resident_read.argtypes=[C.c_void_p, C.POINTER(C.c_uint), C.c_int]
for rep in range(3):
    h=w=96
    img=synth.disk(96,200,50,noise=8,seed=1,h=h,w=w)
    ctx=capi.Context(h,w,1,capi.make_params(tol=0.0))
    ctx.set_option("trace",512)
    ctx.set_image([img]); ctx.init_checkerboard()
    done,nrm=ctx.run(40)
    tr=ctx.get_trace(512)
    buf=(C.c_uint*40)()
    L.cvh_debug_resident_read(ctx._h, buf, 6)
    print("rep",rep,"done",done,"rows",len(tr),"error",buf[0],"flag gens",[buf[2+i] for i in range(6)],"go gens",[buf[8+i] for i in range(6)])
    print("   norms head",np.round(tr[:3,2],3),"rows 32..42",np.round(tr[32:42,2],3) if len(tr)>42 else "", "tail",np.round(tr[-2:,2],3))
    ctx.close()
