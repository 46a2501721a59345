#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_i}
mkdir -p "$O"
step() { local name=$1 lim=$2; shift 2; echo "== $name" | tee -a "$O/session.log"; timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"; local rc=$?; echo "rc=$rc" | tee -a "$O/session.log"; grep -i "linattn_sample\|attn_sample\|total ms\|linear-attention\|passed\|failed\|Error" "$O/$name.out" | cut -c1-200 | tail -n 12; if [ $rc -ge 124 ]; then echo killed; tail -5 "$O/$name.err"; exit $rc; fi; return 0; }
step unet_tests 500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_sizes.py tests/test_gpu_shared_device.py -m gpu -q -x -p no:cacheprovider -k "not test_b_sdvae and not test_d_rk4"
FLOCODER_AMD_LA_HEAD=4 step ops_h4 200 python tools/op_table.py
step ops_h8 200 python tools/op_table.py
FLOCODER_AMD_LA_HEAD=4 step bench_h4 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
step bench_h8 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
FLOCODER_AMD_LA_HEAD=4 step bench_h4b 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
step bench_h8b 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
for n in bench_h4 bench_h8 bench_h4b bench_h8b; do python -c "import json,sys; d=json.loads(open('$O/$n.out').read().strip().splitlines()[-1]); print('$n', d['value'], d['parity_rel_l2'])"; done
