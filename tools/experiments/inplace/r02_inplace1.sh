#!/bin/bash
mkdir -p gpurun_out/inplace
timeout -k 10 400 python tools/inplace_probe.py check 64x160 256x288 512x512 1000x1008 > gpurun_out/inplace/check1.log 2>&1; echo "rc=$?"; cut -c1-330 gpurun_out/inplace/check1.log
