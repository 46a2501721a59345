#!/bin/bash
# Round-4 verification session: the GPU suite, the bench line with and without the host wait in front of the first graph replay.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_g}
mkdir -p "$O"
step() { # name, timeout, command...
    local name=$1 lim=$2; shift 2
    echo "== $name" | tee -a "$O/session.log"
    timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"
    local rc=$?
    echo "rc=$rc" | tee -a "$O/session.log"
    cut -c1-600 "$O/$name.out" | tail -n 6
    if [ $rc -ge 124 ]; then echo "killed: stopping the session" | tee -a "$O/session.log"; tail -5 "$O/$name.err"; exit $rc; fi
    return 0
}
step suite 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8
step bench 600 python bench.py --steps 20 --warmup 5
FLOCODER_AMD_GRAPH_FENCE=0 step bench_nofence 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
step bench_again 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline
echo done | tee -a "$O/session.log"
