import sys; sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
h=w=int(sys.argv[1]) if len(sys.argv)>1 else 4096
planes=[synth.disk(h)]
u0=capi.checkerboard_host(h,w)
def run(opts, steps=1):
    with capi.Context(h,w,1,capi.make_params(tol=0)) as ctx:
        for k,v in opts.items(): ctx.set_option(k,v)
        ctx.set_option("trace", steps)
        ctx.set_image(planes); ctx.set_levelset(u0); ctx.run(steps); return ctx.get_levelset(), ctx.get_trace(steps), ctx.get_means()
ref,trr,mr=run({"kernel":2,"math_mode":1})
print("ref trace",trr, "means after", mr)
for name,opts in (("strict k3",{"kernel":3,"math_mode":1}),("strict k3 sr46 cls0 fin1",{"kernel":3,"math_mode":1,"strip_rows":46,"wave_cls":0,"finalize":1}),("occ4 chain0",{"kernel":3,"wave_occupancy":4,"chain":0}),("occ4 chain0 fin1",{"kernel":3,"wave_occupancy":4,"chain":0,"finalize":1})):
    u,tr,m=run(opts)
    d=np.abs(u-ref); bad=np.nonzero(d>1e-6*np.abs(ref).max())
    print("%-26s max diff %.3e"%(name,d.max()),"bad rows",np.unique(bad[0])[:10],"n",len(np.unique(bad[0])),"cols",np.unique(bad[1])[:6],"n",len(np.unique(bad[1])),"trace",tr,"means",m)
