#!/bin/bash
# tools/raster_prof.sh [tag] -- on the GPU box: rocprofv3 kernel stats + SQ counters of the rasteriser kernels at the BASELINE
# size (tools/softras_microbench.py: B=32, 256x256, 642 v / 1280 f); writes gpurun_out/<tag>_raster_{stats,pmc}.txt
tag=${1:-raster}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag --output-format csv -- python $R/tools/softras_microbench.py > $R/gpurun_out/${tag}_micro.json 2>/dev/null
f=$(ls /tmp/prof_$tag/*/*kernel_stats.csv | head -1)
grep -E "Name|raster_|face_setup|count_pairs" $f | cut -d, -f1-4,6-7 > $R/gpurun_out/${tag}_raster_stats.txt
cd $R
: > gpurun_out/${tag}_raster_pmc.txt
for k in "raster_backward_kernel<1, 1>" "raster_forward_kernel<1, 1>"; do
  echo "== $k" >> gpurun_out/${tag}_raster_pmc.txt
  bash tools/pmc_kernel.sh "$k" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU -- python $R/tools/softras_microbench.py 2>&1 | tail -9 >> gpurun_out/${tag}_raster_pmc.txt
done
cat gpurun_out/${tag}_micro.json gpurun_out/${tag}_raster_stats.txt gpurun_out/${tag}_raster_pmc.txt
