#!/bin/bash
# Round-4 diagnosis session: (1) the whole GPU suite with every library buffer poisoned and fenced, (2) the two-in-flight report, plain and poisoned.
# A step that was killed (rc >= 124) ends the session: no further GPU step after a timeout or a fault.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_a}
mkdir -p "$O"
step() { # name, timeout, command...
    local name=$1 lim=$2; shift 2
    echo "== $name" | tee -a "$O/session.log"
    timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"
    local rc=$?
    echo "rc=$rc" | tee -a "$O/session.log"
    tail -n 6 "$O/$name.out"
    if [ $rc -ge 124 ]; then echo "killed: stopping the session" | tee -a "$O/session.log"; exit $rc; fi
    return 0
}
FLOCODER_AMD_POISON=1 step suite_poison 900 python -m pytest tests -m gpu -q -p no:cacheprovider
FLOCODER_AMD_POISON=1 step diag_poison 400 python tools/inflight_diag.py --rounds 2
step diag_plain 300 python tools/inflight_diag.py --rounds 3
echo done | tee -a "$O/session.log"
