#!/bin/bash
# round 4, session 13: LDS-DMA ring of the 2-pixel kernel (option "wave_dma"): parity, then A/B in one context
set -o pipefail
O=gpurun_out/r4s13; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "two_pixel_kernel_edge_shapes or kernel_flavours_agree_at_4096" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout -k 10 300 python tools/ab_probe.py "wave_dma=0" "wave_dma=1" "wave_dma=1,wave_sync=0" "wave_dma=1,wave_cskew=0" > $O/ab_4096.log 2>&1; cat $O/ab_4096.log
N=2048 timeout -k 10 300 python tools/ab_probe.py "wave_dma=0" "wave_dma=1" > $O/ab_2048.log 2>&1; cat $O/ab_2048.log
N=6144 timeout -k 10 300 python tools/ab_probe.py "wave_dma=0" "wave_dma=1" > $O/ab_6144.log 2>&1; cat $O/ab_6144.log
