#!/bin/bash
# tools/r06/last_check2.sh -- the co-residency file alone after bounding the queued load, then the whole suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/r06y; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_coresidency_gpu.py -x -q -m gpu --durations=12 > $O/coresidency.txt 2>&1; echo "coresidency rc=$? $(grep -E ' passed| failed' $O/coresidency.txt | tail -1)"; grep -A12 slowest $O/coresidency.txt | cut -c1-140
timeout 900 python -m pytest tests/ -x -q -m gpu --durations=8 > $O/suite_g.txt 2>&1; echo "suite_g rc=$? $(grep -E ' passed| failed' $O/suite_g.txt | tail -1)"; grep -A9 slowest $O/suite_g.txt | cut -c1-140
