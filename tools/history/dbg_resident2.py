import sys
sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
for (h,w,steps,tol) in ((96,96,40,0.0),(96,96,40,0.001),(256,256,200,0.0),(2048,2048,100,0.0)):
    img=synth.disk(max(h,w),200,50,noise=8,seed=1,h=h,w=w)
    ctx=capi.Context(h,w,1,capi.make_params(tol=tol))
    ctx.set_option("trace",512)
    print(ctx.launch_info()["kernel"])
    ctx.set_image([img]); ctx.init_checkerboard()
    done,nrm=ctx.run(steps)
    tr=ctx.get_trace(512)
    print((h,w),"run",steps,"tol",tol,"-> done",done,"trace rows",len(tr))
    ctx.enqueue_steps(steps); d2=ctx.sync()
    print("   enqueue",steps,"-> total",d2)
    ctx.close()
