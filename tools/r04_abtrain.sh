#!/bin/bash
# Two builds of the library alternating on ONE box (boxes of the pool differ by ~1.5 %, a build's effect is often smaller): copy the two
# libflocoder_amd.so to ab_libs/lib_head.so and ab_libs/lib_new.so (git-ignored, travels with the snapshot), then
#     gpurun -- 'bash tools/r04_abtrain.sh'
# runs the training tests on the working-tree build and both training benchmarks on each library, twice (FLOCODER_AMD_LIB selects the library).
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
