#!/bin/bash
mkdir -p gpurun_out/persist
timeout -k 10 300 python tools/persist_probe.py check 256x288 1024x1024 > gpurun_out/persist/check1.log 2>&1; echo "rc=$?"; cat gpurun_out/persist/check1.log
timeout -k 10 300 python tools/persist_probe.py time 512x512 1024x1024 2048x2048 4096x4096 > gpurun_out/persist/time1.log 2>&1; echo "rc=$?"; cat gpurun_out/persist/time1.log
