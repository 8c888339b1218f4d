set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
timeout 900 python -m pytest tests/test_project.py tests/test_render_golden.py -q -m gpu > gpurun_out/r05/pytest_fix.txt 2>&1
tail -5 gpurun_out/r05/pytest_fix.txt
bash tools/traffic_run.sh > gpurun_out/r05/traffic_stdout.txt 2>&1
tail -3 gpurun_out/r05/traffic_stdout.txt
ls gpurun_out/
