#!/bin/bash
# tools/r06/pmc_forward_ab.sh -- SQ counters of the two forward rasteriser kernels on the sigma = 1e-3 pass (B = 32): why the pair queue loses
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
echo "== per-face kernel raster_forward_kernel<1, 1, false>" > $O/pmc_forward_ab.txt
timeout 280 bash tools/pmc_kernel.sh "raster_forward_kernel<1, 1, false>" $C -- python $R/tools/softras_microbench.py >> $O/pmc_forward_ab.txt 2>&1
echo "== pair-queue kernel raster_forward_pq_kernel<false>" >> $O/pmc_forward_ab.txt
SCP_RASTER_FWD=pq timeout 280 bash tools/pmc_kernel.sh "raster_forward_pq_kernel<false>" $C -- python $R/tools/softras_microbench.py >> $O/pmc_forward_ab.txt 2>&1
cat $O/pmc_forward_ab.txt
