import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
L=capi.lib()
L.cvh_debug_resident_read.argtypes=[C.c_void_p, C.POINTER(C.c_uint), C.c_int]
def rd(ctx):
    buf=(C.c_uint*40)(); L.cvh_debug_resident_read(ctx._h, buf, 2)
    return "err %d t_first(arg) %d nit %d steps_done(read) %d flag gen %d go gen %d"%(buf[0], buf[1]&0xfff, (buf[1]>>12)&0xfff, buf[1]>>24, buf[2], buf[4])
for rep,(pm,li) in enumerate(((0,1),(1,0),(0,0),(1,1),(1,0))):
    h=w=96
    img=synth.disk(96,200,50,noise=8,seed=1,h=h,w=w)
    ctx=capi.Context(h,w,1,capi.make_params(tol=0.0))
    ctx.set_option("trace",512)
    if li: ctx.launch_info()
    ctx.set_image([img]); ctx.init_checkerboard()
    if pm: ctx.perona_malik(30.0,0.25,5.0)
    done,nrm=ctx.run(40)
    print("rep",rep,"pm",pm,"launch_info first",li,"-> done",done, rd(ctx))
    ctx.enqueue_steps(100); d2=ctx.sync(); print("    enqueue 100 ->",d2[0], rd(ctx))
    ctx.close()
