#!/bin/sh
# tools/probes/build_gemm_variants.sh -- ablation builds of csrc/vit_gemm.hip (SCP_PROBE_* switches of gemm_core_split.h) linked with
# the shipped objects into tools/probes/libscp_<variant>.bin; load one with SCP_HIP_LIB=... (timing only, results are wrong by design)
set -e
cd "$(dirname "$0")/../../self-corr-pose_amd"
python build.py > /dev/null
for v in NO_MFMA NO_DMA NO_SPLIT; do
  (/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I ../include -I csrc -DSCP_PROBE_$v -c csrc/vit_gemm.hip -o /tmp/vit_gemm_$v.o &&
   /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I ../include -I csrc -DSCP_PROBE_$v -c csrc/conv_igemm.hip -o /tmp/conv_igemm_$v.o &&
   /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v "vit_gemm.o\|conv_igemm.o") /tmp/vit_gemm_$v.o /tmp/conv_igemm_$v.o -o ../tools/probes/libscp_$v.bin) &
done
wait
ls -la ../tools/probes/*.bin
