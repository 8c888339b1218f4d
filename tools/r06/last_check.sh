#!/bin/bash
# tools/r06/last_check.sh -- after the co-residency engine changes: that file alone with durations, then the whole suite once more
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06x; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_coresidency_gpu.py tests/test_imgops.py -x -q -m gpu --durations=20 > $O/coresidency.txt 2>&1; echo "coresidency rc=$? $(grep -E ' passed| failed' $O/coresidency.txt | tail -1)"; grep -A20 slowest $O/coresidency.txt | cut -c1-140
timeout 900 python -m pytest tests/ -x -q -m gpu --durations=8 > $O/suite_f.txt 2>&1; echo "suite_f rc=$? $(grep -E ' passed| failed' $O/suite_f.txt | tail -1)"; grep -A9 slowest $O/suite_f.txt | cut -c1-140
