import sys; sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
from oracle import cv_oracle as O
h,w=int(sys.argv[1]),int(sys.argv[2])
rng=np.random.default_rng(3)
planes=[rng.integers(0,256,size=(h,w),dtype=np.uint8) for _ in range(3)]
u0=O.checkerboard(h,w)
pk=dict(tol=0,lambda1=[1,.8,.5],lambda2=[.7,.5,1])
for math in (1,2):
  for kern in (2,3):
    uc=u0.copy(); nrm,c1,c2=O.csv_step(planes,uc,O.make_params(**pk))
    with capi.Context(h,w,3,capi.make_params(**pk)) as ctx:
        ctx.set_option("math_mode",math); ctx.set_option("kernel",kern); ctx.set_option("trace",2)
        ctx.set_image(planes); ctx.set_levelset(u0); ctx.run(1); ug=ctx.get_levelset(); tr=ctx.get_trace(1)
    d=np.abs(ug-uc); i,j=np.unravel_index(d.argmax(),d.shape)
    print("math",math,"kernel",kern,"max err %.3e at (%d,%d) rel %.2e; bad cols:"%(d.max(),i,j,d.max()/np.abs(uc).max()), np.unique(np.nonzero(d>1e-9*np.abs(uc).max())[1])[:12], "trace", tr[0][:3], "oracle c1", c1)
