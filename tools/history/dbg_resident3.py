import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
L=capi.lib()
L.cvh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
for rep in range(3):
    h=w=96
    img=synth.disk(96,200,50,noise=8,seed=1,h=h,w=w)
    ctx=capi.Context(h,w,1,capi.make_params(tol=0.0))
    ctx.set_option("trace",512); ctx.set_option("debug_times",1)
    ctx.set_image([img]); ctx.init_checkerboard()
    done,nrm=ctx.run(40)
    buf=np.zeros(20000,dtype=np.uint64); words=C.c_long(0); nb=C.c_int(0)
    L.cvh_debug_read(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(words), C.byref(nb))
    w_=buf[:6*12].reshape(6,12)
    print("rep",rep,"done",done,"t_first per tile",w_[:,9],"nit",w_[:,10])
    ctx.close()
