set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
timeout 1500 python -m pytest tests/test_project.py tests/test_step_gpu.py tests/test_trainer_host.py tests/test_parallel.py tests/test_coresidency_gpu.py -q -m gpu > gpurun_out/r05/pytest_adamw.txt 2>&1
tail -8 gpurun_out/r05/pytest_adamw.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('flat adamw', d['ms_per_step'], d['config']['vit_lookahead']['unpipelined_ms_per_step'], d['gradients'], d['roofline']['frac'], d['roofline']['exclusive_device']['frac'])"
SCP_ADAMW=torch timeout 600 python bench.py --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('torch adamw', d['ms_per_step'], d['config']['vit_lookahead']['unpipelined_ms_per_step'], d['gradients'])"
done
