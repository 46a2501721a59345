#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
  for v in head new; do
    echo "== $v (run $rep)"
    FLOCODER_AMD_LIB=$PWD/ab_libs/lib_$v.so timeout -k 10 200 python tools/bench_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stl_sd', d['ms_per_step'])"
    FLOCODER_AMD_LIB=$PWD/ab_libs/lib_$v.so timeout -k 10 200 python tools/bench_train.py --dim 32 --hw 32 --batch 64 --classes 102 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flowers', d['ms_per_step'])"
  done
done
