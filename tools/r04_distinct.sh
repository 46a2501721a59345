#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04_f}
mkdir -p "$O"
step() { local name=$1 lim=$2; shift 2; echo "== $name" | tee -a "$O/session.log"; timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"; local rc=$?; echo "rc=$rc" | tee -a "$O/session.log"; cut -c1-700 "$O/$name.out" | tail -n 4; if [ $rc -ge 124 ]; then echo killed; tail -5 "$O/$name.err"; exit $rc; fi; return 0; }
step base 200 python tools/inflight_distinct.py --in-flight 1
FLOCODER_AMD_GRAPH_FENCE=1 step fence_hostsync 200 python tools/inflight_distinct.py --in-flight 1
FLOCODER_AMD_GRAPH_FENCE=2 step fence_event 200 python tools/inflight_distinct.py --in-flight 1
FLOCODER_AMD_GRAPH_STEPS=1 step one_interval_per_graph 200 python tools/inflight_distinct.py --in-flight 1
echo done
