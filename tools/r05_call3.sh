set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
# (a) exact (mangled) names of every kernel a step launches, overlapped schedule
rm -rf /tmp/tr_names
SCP_STREAMS=overlap rocprofv3 --kernel-trace -M -d /tmp/tr_names --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $R/gpurun_out/r05/names_bench.json 2>/dev/null
f=$(ls /tmp/tr_names/*/*kernel_trace.csv | head -1)
python - "$f" > $R/gpurun_out/r05/step_kernel_names.txt <<'PY'
import csv, sys, collections
c = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    c[row["Kernel_Name"]] += 1
for k, n in sorted(c.items()):
    print(n, k)
PY
wc -l $R/gpurun_out/r05/step_kernel_names.txt
cd $R
# (b) the screen as a table
SCP_STREAMS=overlap timeout 900 python tests/coresidency.py 60 > gpurun_out/r05/screen_overlap.txt 2>&1
tail -20 gpurun_out/r05/screen_overlap.txt
# (c) the screen as tests
SCP_STREAMS=overlap SCP_SCREEN_PASSES=100 timeout 1200 python -m pytest tests/test_coresidency_gpu.py -x -q > gpurun_out/r05/screen_pytest.txt 2>&1
tail -15 gpurun_out/r05/screen_pytest.txt
