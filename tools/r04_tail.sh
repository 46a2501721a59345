#!/bin/bash
# the Block-closing tail: stamps, op table, parity tests, headline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_tail}
mkdir -p "$O"
step() { local name=$1 lim=$2; shift 2; echo "== $name" | tee -a "$O/session.log"; timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"; local rc=$?; echo "rc=$rc" | tee -a "$O/session.log"; grep -i "tail:\|+fin \|total ms\|passed\|failed\|Error" "$O/$name.out" | cut -c1-330 | tail -n 30; if [ $rc -ge 124 ]; then echo killed; tail -5 "$O/$name.err"; exit $rc; fi; return 0; }
step stamps 300 python tools/fin_stamps.py
step unet_tests 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_sizes.py tests/test_gpu_shared_device.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "not test_b_sdvae and not test_d_rk4"
step ops 200 python tools/op_table.py
step bench1 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
step bench2 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
for n in bench1 bench2; do python -c "import json,sys; d=json.loads(open('$O/$n.out').read().strip().splitlines()[-1]); print('$n', d['value'], d['parity_rel_l2'])"; done
