#!/bin/bash
# Builds the libraries profiles/step_timeline.sh runs: the ablation objects of csrc/Makefile with
# kernels_fused.hip compiled once per reading mask (-DRDAMD_ABL_STAMPS=<mask>, kernels_fused.hip) ->
# profiles/tmp_libs/stamps_<mask>.so.  Runs in the CPU container (hipcc cross-compiles).
# usage: profiles/step_timeline_build.sh [masks...]      (default: 31 1 3 5 9 17)
set -e
cd "$(dirname "$0")/.."
MASKS=${@:-31 1 3 5 9 17}
CS=root_digger_amd/csrc
make -s -C $CS ablation -j8 > /tmp/abl_build.log 2>&1 || { tail -20 /tmp/abl_build.log; exit 1; }
mkdir -p profiles/tmp_libs $CS/build/abl_st
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-c99-designator -mllvm -amdgpu-mfma-vgpr-form -DRDAMD_ABLATION"
OTHERS=$(ls $CS/build/abl/*.o | grep -v kernels_fused.hip.o)
for m in $MASKS; do
  ( /opt/rocm/bin/hipcc $FLAGS -DRDAMD_ABL_STAMPS=$m -c $CS/kernels_fused.hip -o $CS/build/abl_st/kf_$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o profiles/tmp_libs/stamps_$m.so $OTHERS $CS/build/abl_st/kf_$m.o &&
    echo "built profiles/tmp_libs/stamps_$m.so" ) &
done
wait
