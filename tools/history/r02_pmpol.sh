#!/bin/bash
mkdir -p gpurun_out/aux
: > gpurun_out/aux/pmpol.log
for n in 2048 4096 6144; do
  N=$n python tools/pm_ab.py "wave_pol=1" "wave_pol=0" "wave_pol=-1" >> gpurun_out/aux/pmpol.log 2>&1
done
cat gpurun_out/aux/pmpol.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "perona or pm_ or flavours or config4 or pipeline" > gpurun_out/aux/pytest_pm.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/aux/pytest_pm.log
