cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_lh
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/t_lh.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/b_lh.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lh -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_lh.log 2>&1
