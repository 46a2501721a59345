#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_h}
mkdir -p "$O"
step() { local name=$1 lim=$2; shift 2; echo "== $name" | tee -a "$O/session.log"; timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"; local rc=$?; echo "rc=$rc" | tee -a "$O/session.log"; grep -i "linattn_fused\|total ms\|linear-attention\|passed\|failed" "$O/$name.out" | cut -c1-200 | tail -n 12; if [ $rc -ge 124 ]; then echo killed; tail -5 "$O/$name.err"; exit $rc; fi; return 0; }
step unet_tests 400 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench_sizes.py -m gpu -q -x -p no:cacheprovider -k "not test_b_sdvae"
FLOCODER_AMD_LA_CTX_WAVES=4 step ops_w4 200 python tools/op_table.py
step ops_w8 200 python tools/op_table.py
FLOCODER_AMD_LA_APPLY_GX=full step ops_w8_gxfull 200 python tools/op_table.py
FLOCODER_AMD_LA_CTX_WAVES=4 step bench_w4 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
step bench_w8 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
FLOCODER_AMD_LA_APPLY_GX=full step bench_w8_gxfull 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
FLOCODER_AMD_LA_CTX_WAVES=4 step bench_w4b 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
for n in bench_w4 bench_w8 bench_w8_gxfull bench_w4b; do python -c "import json,sys; d=json.loads(open('$O/$n.out').read().strip().splitlines()[-1]); print('$n', d['value'], d['parity_rel_l2'])"; done
