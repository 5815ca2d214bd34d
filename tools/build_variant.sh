#!/bin/bash
# tools/build_variant.sh NAME "-DMACRO=..."  -> chan_vese_amd/csrc/variants/NAME/libchanvese_hip.so (A/B builds, git-ignored)
# Only ONE kernel file (FILE=csv_wave2_kernel by default, e.g. FILE=csv_resident_kernel) is rebuilt with the extra flags; the other
# objects are the ones of the default build.
set -e
cd "$(dirname "$0")/../chan_vese_amd/csrc"
name=$1; shift
mkdir -p variants/$name
FILE=${FILE:-csv_wave2_kernel}
objs=""
for f in api csv_kernels csv_wave_kernel csv_wave2_kernel csv_resident_kernel pm_kernels pm_wave_k2_kernel pm_resident_kernel chain_kernels misc_kernels; do
  [ $f = $FILE ] || objs="$objs $f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-function "$@" -c $FILE.hip -o variants/$name/$FILE.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/$name/libchanvese_hip.so $objs variants/$name/$FILE.o
echo built variants/$name
