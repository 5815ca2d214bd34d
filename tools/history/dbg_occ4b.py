import sys; sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
h=w=2048
planes=[synth.disk(h)]
cb=capi.checkerboard_host(h,w)
for name,u0 in (("far start (|u|=100)", 100.0*np.where(cb==0,1.0,cb)), ("near start (checkerboard)", cb), ("mixed (|u|=31..33 stripes)", np.where((np.arange(h)[:,None]//64)%2==0, 31.0, 33.0)*np.where(cb==0,1.0,cb))):
    def run(opts, steps=1):
        with capi.Context(h,w,1,capi.make_params(tol=0)) as ctx:
            for k,v in opts.items(): ctx.set_option(k,v)
            ctx.set_image(planes); ctx.set_levelset(u0); ctx.run(steps); return ctx.get_levelset(), ctx.get_means()
    ref,mr=run({"kernel":2})
    for nm,opts in (("k3",{"kernel":3}),("k3occ4",{"kernel":3,"wave_occupancy":4})):
        u,m=run(opts)
        d=np.abs(u-ref); bad=np.nonzero(d>1e-9*np.abs(ref).max())
        lanes=np.unique(((np.unique(bad[1])%126)+2)//2) if len(bad[1]) else []
        print("%-28s %-7s max diff %.3e bad rows n %d cols n %d lanes %s means diff %.2e %.2e"%(name,nm,d.max(),len(np.unique(bad[0])),len(np.unique(bad[1])),list(lanes)[:20],abs(m[0][0]-mr[0][0]),abs(m[1][0]-mr[1][0])))
