import sys; sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
for (h,w) in [(512,528),(1024,1024),(2048,2048),(4096,4096)]:
    planes=[synth.disk(min(h,w),180,40,h=h,w=w),synth.disk(min(h,w),200,60,h=h,w=w),synth.disk(min(h,w),60,200,h=h,w=w)]
    u0=capi.checkerboard_host(h,w)
    pk=dict(tol=0,lambda1=[1,1,.5],lambda2=[1,.5,1])
    res={}
    for kern,extra in ((2,{}),(3,{}),(3,{"wave_cskew":0}),(3,{"chain":0})):
        with capi.Context(h,w,3,capi.make_params(**pk)) as ctx:
            ctx.set_option("kernel",kern); ctx.set_option("trace",2)
            for k,v in extra.items(): ctx.set_option(k,v)
            ctx.set_image(planes); ctx.set_levelset(u0); ctx.run(2); res[(kern,str(extra))]=(ctx.get_levelset(),ctx.get_trace(2))
    ref=res[(2,'{}')][0]
    for key,(u,tr) in res.items():
        d=np.abs(u-ref); bad=np.nonzero(d>1e-9*np.abs(ref).max())
        print(h,w,key,"max diff vs kernel2 %.3e"%d.max(),"bad rows",np.unique(bad[0])[:8],"n",len(np.unique(bad[0])),"bad cols",np.unique(bad[1])[:8],"n",len(np.unique(bad[1])),"trace",tr[1][:3])
