#!/bin/bash
# tools/kernel_regs.sh <file.hip> [filter] -- VGPR / SGPR / LDS / scratch of every kernel of one csrc file (cross-compiles for gfx950)
f=$1; pat=${2:-.}
d=$(cd "$(dirname "$0")/../self-corr-pose_amd" && pwd)
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I$d/csrc -I$d/../include --cuda-device-only -S $d/csrc/$f -o $tmp/k.s 2>/dev/null || exit 1
awk '/^[ \t]*\.amdhsa_kernel/ {k=$2} /\.amdhsa_next_free_vgpr/ {v=$2} /\.amdhsa_next_free_sgpr/ {s=$2} /\.amdhsa_group_segment_fixed_size/ {l=$2} /\.amdhsa_private_segment_fixed_size/ {p=$2} /^[ \t]*\.end_amdhsa_kernel/ {print v, s, l, p, k}' $tmp/k.s | while read v s l p k; do echo "vgpr $v sgpr $s lds $l scratch $p  $(echo $k | c++filt | cut -c1-150)"; done | grep -E "$pat"
rm -rf $tmp
