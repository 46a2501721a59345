#!/bin/bash
# Round-4 localisation session: two replicas driven by two host threads (tools/race_hunt.py), default build switches and knock-outs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_b}
mkdir -p "$O"
step() { # name, timeout, command...
    local name=$1 lim=$2; shift 2
    echo "== $name" | tee -a "$O/session.log"
    timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"
    local rc=$?
    echo "rc=$rc" | tee -a "$O/session.log"
    cut -c1-1500 "$O/$name.out" | tail -n 8
    if [ $rc -ge 124 ]; then echo "killed: stopping the session" | tee -a "$O/session.log"; tail -5 "$O/$name.err"; exit $rc; fi
    return 0
}
step hunt_default 300 python tools/race_hunt.py --iters 6000
FLOCODER_AMD_NO_W4=1 step hunt_no_w4 300 python tools/race_hunt.py --iters 6000
FLOCODER_AMD_UPS_FOLD=0 step hunt_no_fold 300 python tools/race_hunt.py --iters 6000
FLOCODER_AMD_LEAN_KERNELS=0 step hunt_no_lean 300 python tools/race_hunt.py --iters 6000
echo done | tee -a "$O/session.log"
