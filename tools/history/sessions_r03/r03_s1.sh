#!/bin/bash
# round 3, session 1: the store-hazard probe, then the whole GPU suite with durations
set -o pipefail
mkdir -p gpurun_out/r3s1
timeout -k 10 300 tools/store_hazard_probe 3 > gpurun_out/r3s1/hazard_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r3s1/hazard_probe.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r3s1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3s1/pytest.log
tail -5 gpurun_out/r3s1/pytest.log
