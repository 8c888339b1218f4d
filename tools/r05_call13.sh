R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
for mode in online read; do for i in 1 2 3; do
  SCP_GEMM_TUNING=$mode timeout 600 python -m pytest "tests/test_step_gpu.py::test_trainer_step_runs_and_updates" tests/test_parallel.py -q -m gpu > gpurun_out/r05/abort_$mode$i.txt 2>&1
  echo "$mode run $i: $(grep -c 'Fatal Python' gpurun_out/r05/abort_$mode$i.txt) aborts; $(tail -1 gpurun_out/r05/abort_$mode$i.txt | cut -c1-100)"
done; done
