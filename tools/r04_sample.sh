#!/bin/bash
# the one-workgroup-per-sample kernel: its golden-vector tests, the config-5 inpainting bench with and without it, the per-step stamps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_s}
mkdir -p "$O"
step() { local name=$1 lim=$2; shift 2; echo "== $name" | tee -a "$O/session.log"; timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"; local rc=$?; echo "rc=$rc" | tee -a "$O/session.log"; grep -i "unet_sample\|passed\|failed\|Error\|us_per_evaluation" "$O/$name.out" "$O/$name.err" | cut -c1-400 | tail -n 8; if [ $rc -ge 124 ]; then echo killed; tail -5 "$O/$name.err"; exit $rc; fi; return 0; }
export FLOCODER_AMD_SAMPLE_KERNEL_DEBUG=1
FLOCODER_AMD_SAMPLE_KERNEL=1 step tests 400 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -p no:cacheprovider -k "d8mask or mask_cond_sampling or one_workgroup"
FLOCODER_AMD_SAMPLE_KERNEL=1 step inpaint_sample 300 python tools/bench_inpaint.py
step inpaint_plan 300 python tools/bench_inpaint.py
FLOCODER_AMD_SAMPLE_KERNEL=1 step stamps 200 python tools/sample_kernel_stamps.py
tail -n 30 "$O/stamps.out"
