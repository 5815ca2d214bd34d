import sys; sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
h=w=int(sys.argv[1]) if len(sys.argv)>1 else 4096
planes=[synth.disk(h)]
u0=capi.checkerboard_host(h,w)
def run(opts, steps=3):
    with capi.Context(h,w,1,capi.make_params(tol=0)) as ctx:
        for k,v in opts.items(): ctx.set_option(k,v)
        ctx.set_image(planes); ctx.set_levelset(u0); ctx.run(steps); return ctx.get_levelset()
ref=run({"kernel":2,"math_mode":1}); reff=run({"kernel":2})
for name,opts,r in (("strict",{"kernel":3,"math_mode":1},ref),("strict cls0",{"kernel":3,"math_mode":1,"wave_cls":0},ref),("strict cskew0",{"kernel":3,"math_mode":1,"wave_cskew":0},ref),
                    ("strict sr46",{"kernel":3,"math_mode":1,"strip_rows":46},ref),("strict sr46 cls0",{"kernel":3,"math_mode":1,"strip_rows":46,"wave_cls":0},ref),
                    ("occ4",{"kernel":3,"wave_occupancy":4},reff),("occ4 cls0",{"kernel":3,"wave_occupancy":4,"wave_cls":0},reff),("occ4 chain0",{"kernel":3,"wave_occupancy":4,"chain":0},reff),
                    ("occ4 sr46",{"kernel":3,"wave_occupancy":4,"strip_rows":46},reff),("fast sr68",{"kernel":3,"strip_rows":68},reff),("fast sr100 cls0",{"kernel":3,"strip_rows":100,"wave_cls":0},reff),("fast graph0",{"kernel":3,"graph":0},reff)):
    d=np.abs(run(opts)-r); bad=np.nonzero(d>1e-9*np.abs(r).max())
    print("%-18s max diff %.3e"%(name,d.max()),"bad rows",np.unique(bad[0])[:6],"n",len(np.unique(bad[0])),"cols",np.unique(bad[1])[:6],"n",len(np.unique(bad[1])))
