#!/bin/bash
mkdir -p gpurun_out/s14
python tools/ab_probe.py wave_newton=0 wave_newton=1 > gpurun_out/s14/ab_newton.log 2>&1; cat gpurun_out/s14/ab_newton.log
N=2048 python tools/ab_probe.py wave_newton=0 wave_newton=1 > gpurun_out/s14/ab_newton2048.log 2>&1; cat gpurun_out/s14/ab_newton2048.log
python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np
from chan_vese_amd import capi, synth
from oracle import cv_oracle as O
# accuracy of the Newton flavour against the oracle: 512^2 disk, 100 iterations, and a noisy 300x512
for name,img in (("disk512", synth.disk(512)), ("noisy", synth.disk(300, 190, 60, noise=24, seed=4, h=300, w=512))):
    h,w=img.shape; u0=O.checkerboard(h,w)
    uc,_,_,trc=O.csv_run([img],u0,O.make_params(tol=0),100)
    for nt in (0,1):
        with capi.Context(h,w,1,capi.make_params(tol=0.0)) as ctx:
            ctx.set_option("kernel",3); ctx.set_option("wave_newton",nt); ctx.set_option("trace",100)
            ctx.set_image([img]); ctx.set_levelset(u0); ctx.run(10); u10=ctx.get_levelset(); ctx.run(90); u=ctx.get_levelset(); tr=ctx.get_trace(100)
        u10c,_,_,_=O.csv_run([img],u0,O.make_params(tol=0),10)
        print(name,"newton",nt,"rel err @10 %.2e @100 %.2e  trace rel max %.2e"%(np.abs(u10-u10c).max()/np.abs(u10c).max(), np.abs(u-uc).max()/np.abs(uc).max(), np.abs(tr/trc-1).max()))
PY
