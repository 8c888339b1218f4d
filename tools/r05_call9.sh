set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05; cd $R
timeout 600 python tools/conv_bench.py > gpurun_out/r05/conv_layers_3stage.txt 2>&1
SCP_HIP_LIB=$R/self-corr-pose_amd/lib/libscp_hip_conv2.so timeout 600 python tools/conv_bench.py > gpurun_out/r05/conv_layers_2stage.txt 2>&1
tail -22 gpurun_out/r05/conv_layers_3stage.txt | cut -c1-200
tail -3 gpurun_out/r05/conv_layers_2stage.txt | cut -c1-200
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fused_conv.py tests/test_render_golden.py -q -m gpu 2>&1 | tail -4
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('3stage', d['ms_per_step'], d['config']['vit_lookahead']['unpipelined_ms_per_step'])"
SCP_HIP_LIB=$R/self-corr-pose_amd/lib/libscp_hip_conv2.so timeout 600 python bench.py --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2stage', d['ms_per_step'], d['config']['vit_lookahead']['unpipelined_ms_per_step'])"
done
