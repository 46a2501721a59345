cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_gpu_unet.py tests/test_gpu_kernels.py tests/test_gpu_vae.py tests/test_gpu_vqvae.py -m gpu -x -q > gpurun_out/t_lh.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/b_lh.log 2>&1
PYTHONPATH=. timeout -k 10 400 python tools/bench_decode.py > gpurun_out/codec_x.log 2>&1
