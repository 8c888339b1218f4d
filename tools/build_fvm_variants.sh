#!/bin/bash
# tools/build_fvm_variants.sh -- probe builds of csrc/corr_fused.hip / csrc/corr_pp.hip (occupancy hints of the correspondence
# kernels) linked with the objects of the regular build into self-corr-pose_amd/lib/variants/libscp_<name>.so; time them with
#   SCP_HIP_LIB=<that .so> python tools/fvm_bench.py | tools/pp_bench.py
set -e
d=$(cd "$(dirname "$0")/../self-corr-pose_amd" && pwd)
mkdir -p $d/lib/variants
build() {   # name, source file, extra flags
    name=$1; src=$2; shift; shift
    objs=$(ls $d/build/*.o | grep -v "/${src%.hip}.o")
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I $d/../include -I $d/csrc "$@" -c $d/csrc/$src -o /tmp/var_$name.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/var_$name.o -o $d/lib/variants/libscp_$name.so
    echo built $name
}
build fvm_i2m2 corr_fused.hip -DFVM_IMG_WAVES=2 -DFVM_MESH_WAVES=2 &
build fvm_i3m3 corr_fused.hip -DFVM_IMG_WAVES=3 -DFVM_MESH_WAVES=3 &
build pp_f3b3 corr_pp.hip -DPP_FWD_WAVES=3 -DPP_BWD_WAVES=3 &
build pp_f3b2 corr_pp.hip -DPP_FWD_WAVES=3 -DPP_BWD_WAVES=2 &
wait
