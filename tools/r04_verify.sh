#!/bin/bash
# Round-4 verification session on the final build: race screen, two-in-flight report, the GPU suite with every buffer poisoned and fenced.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_l}
mkdir -p "$O"
step() { # name, timeout, command...
    local name=$1 lim=$2; shift 2
    echo "== $name" | tee -a "$O/session.log"
    timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"
    local rc=$?
    echo "rc=$rc" | tee -a "$O/session.log"
    cut -c1-400 "$O/$name.out" | tail -n 5
    if [ $rc -ge 124 ]; then echo "killed: stopping the session" | tee -a "$O/session.log"; tail -5 "$O/$name.err"; exit $rc; fi
    return 0
}
step hunt 300 python tools/race_hunt.py --iters 8000
FLOCODER_AMD_POISON=1 step suite_poison 1100 python -m pytest tests -m gpu -q -p no:cacheprovider
echo done | tee -a "$O/session.log"
