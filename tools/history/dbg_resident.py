import ctypes as C, sys
sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi
from oracle import cv_oracle as O
L=capi.lib()
L.cvh_debug_resident_read.argtypes=[C.c_void_p, C.POINTER(C.c_uint), C.c_int]
for (h,w) in ((32,256),(16,256),(32,128)):
  for steps in (1,2,3):
    rng=np.random.default_rng(1); img=rng.integers(0,256,size=(h,w),dtype=np.uint8)
    u0=O.checkerboard(h,w)
    ctx=capi.Context(h,w,1,capi.make_params(tol=0))
    ctx.set_option("resident",1); ctx.set_option("trace",8)
    info=ctx.launch_info()
    ctx.set_image([img]); ctx.set_levelset(u0)
    try:
        done,nrm=ctx.run(steps)
    except Exception as e:
        print("EXC",e); done=-1
    buf=(C.c_uint*40)()
    L.cvh_debug_resident_read(ctx._h, buf, 8)
    print((h,w),"steps",steps,"done",done,"grid",info["grid"],"tiles",info["tiles_y"],info["tiles_x"],"arrive",buf[0],"error",buf[1],"go",[hex(buf[2+i]) for i in range(int(info["grid"]))], "trace", ctx.get_trace(8)[:, -1])
    ctx.close()
