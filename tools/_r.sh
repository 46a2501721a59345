cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_gpu_unet.py tests/test_gpu_train.py -m gpu -x -q > gpurun_out/t_lh.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/b_lh.log 2>&1
