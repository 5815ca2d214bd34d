#!/bin/bash
# device-side image sums / checkerboard: parity subset + host-visible time per image
mkdir -p gpurun_out/e2e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "checkerboard or stop_condition or stop_rule or pm_then_csv or pipeline or chain_mode" > gpurun_out/e2e/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/e2e/pytest.log
python tools/e2e_probe.py 4096 1 500 > gpurun_out/e2e/e2e.log 2>&1; python tools/e2e_probe.py 4096 3 300 >> gpurun_out/e2e/e2e.log 2>&1; cat gpurun_out/e2e/e2e.log
