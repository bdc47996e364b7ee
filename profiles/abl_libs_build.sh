#!/bin/bash
# ablation libraries that differ in kernels_fused.hip's switches only: profiles/tmp_libs/<name>.so
# usage: profiles/abl_libs_build.sh name1="-DFLAG ..." name2="..."      (name=""  : the plain ablation build)
set -e
cd "$(dirname "$0")/.."
CS=root_digger_amd/csrc
make -s -C $CS ablation -j8 > /tmp/abl_build.log 2>&1 || { tail -20 /tmp/abl_build.log; exit 1; }
mkdir -p profiles/tmp_libs $CS/build/abl_st
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-c99-designator -mllvm -amdgpu-mfma-vgpr-form -DRDAMD_ABLATION"
FILE=${ABL_FILE:-kernels_fused.hip}
OTHERS=$(ls $CS/build/abl/*.o | grep -v $FILE.o)
for spec in "$@"; do
  name=${spec%%=*}; extra=${spec#*=}
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c $CS/$FILE -o $CS/build/abl_st/$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o profiles/tmp_libs/$name.so $OTHERS $CS/build/abl_st/$name.o &&
    echo "built profiles/tmp_libs/$name.so" ) &
done
wait
