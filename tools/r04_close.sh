#!/bin/bash
# the fused close of the n >= 256 linear attention (la_apply normalises and adds x itself): parity tests, the op table, the headline with and without
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_close}
mkdir -p "$O"
step() { local name=$1 lim=$2; shift 2; echo "== $name" | tee -a "$O/session.log"; timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"; local rc=$?; echo "rc=$rc" | tee -a "$O/session.log"; grep -i "linattn_fused\|finalize  \|total ms\|linear-attention\|passed\|failed\|Error" "$O/$name.out" | cut -c1-200 | tail -n 14; if [ $rc -ge 124 ]; then echo killed; tail -5 "$O/$name.err"; exit $rc; fi; return 0; }
step unet_tests 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_sizes.py tests/test_gpu_shared_device.py tests/test_gpu_fused_tail.py -m gpu -q -x -p no:cacheprovider -k "not test_b_sdvae and not test_d_rk4"
step ops 200 python tools/op_table.py
FLOCODER_AMD_LA_CLOSE=0 step bench_off 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
step bench_on 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
FLOCODER_AMD_LA_CLOSE=0 step bench_off2 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
step bench_on2 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
for n in bench_off bench_on bench_off2 bench_on2; do python -c "import json,sys; d=json.loads(open('$O/$n.out').read().strip().splitlines()[-1]); print('$n', d['value'], d['parity_rel_l2'])"; done
