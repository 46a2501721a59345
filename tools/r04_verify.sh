#!/bin/bash
# Round-4 verification session: race screen, two-in-flight report, the GPU suite, the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_c}
mkdir -p "$O"
step() { # name, timeout, command...
    local name=$1 lim=$2; shift 2
    echo "== $name" | tee -a "$O/session.log"
    timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"
    local rc=$?
    echo "rc=$rc" | tee -a "$O/session.log"
    cut -c1-1200 "$O/$name.out" | tail -n 6
    if [ $rc -ge 124 ]; then echo "killed: stopping the session" | tee -a "$O/session.log"; tail -5 "$O/$name.err"; exit $rc; fi
    return 0
}
step hunt 300 python tools/race_hunt.py --iters 8000
step diag 300 python tools/inflight_diag.py --rounds 3
step suite 1100 python -m pytest tests -m gpu -q -p no:cacheprovider
step bench 600 python bench.py --steps 20 --warmup 5
echo done | tee -a "$O/session.log"
