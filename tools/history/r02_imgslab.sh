#!/bin/bash
# one image-load instruction per group for all channels (planes in one slab): parity subset, then C=3 timing old / new library alternately
mkdir -p gpurun_out/imgslab
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "three_channel or edge or small_shapes or golden or chain_mode or flavours or kernel" > gpurun_out/imgslab/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/imgslab/pytest.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "3ch or config3 or three" > gpurun_out/imgslab/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/imgslab/pytest2.log
for r in 1 2; do
  CHANVESE_HIP_LIB=$PWD/tools/_old_libchanvese_hip.so C=3 REPS=3 python tools/ab_probe.py "chain=1" >> gpurun_out/imgslab/ab.log 2>&1
  C=3 REPS=3 python tools/ab_probe.py "chain=1" >> gpurun_out/imgslab/ab.log 2>&1
done
cat gpurun_out/imgslab/ab.log
