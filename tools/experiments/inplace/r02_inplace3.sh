#!/bin/bash
mkdir -p gpurun_out/inplace
V=chan_vese_amd/csrc/variants
OPTS=inplace=1 REPS=3 timeout -k 10 300 python tools/ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/ip_nostop/libchanvese_hip.so $V/ip_nopub/libchanvese_hip.so $V/ip_nofirst/libchanvese_hip.so $V/ip_none/libchanvese_hip.so > gpurun_out/inplace/abl4096.log 2>&1; cat gpurun_out/inplace/abl4096.log
N=2048 OPTS=inplace=1 REPS=3 timeout -k 10 300 python tools/ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/ip_nostop/libchanvese_hip.so $V/ip_nopub/libchanvese_hip.so $V/ip_nofirst/libchanvese_hip.so $V/ip_none/libchanvese_hip.so > gpurun_out/inplace/abl2048.log 2>&1; cat gpurun_out/inplace/abl2048.log
