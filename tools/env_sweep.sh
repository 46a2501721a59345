#!/bin/bash
# Runtime-knob sweep for the headline sampler (round 3): each line = one `bench.py` run (no roofline / secondary / CPU legs) under one
# environment setting of the HIP runtime / of this library; prints samples/s and ms per step.
# Run on the GPU box:  bash tools/env_sweep.sh > gpurun_out/<tag>/env_sweep.txt
run() {
  v=$(env FLOCODER_AMD_KEEP_ENV=1 "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])" 2>/dev/null)
  echo "$* -> $v"
}
run X=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_DIRECT_DISPATCH=0 ROC_SYSTEM_SCOPE_SIGNAL=0
run HSA_ENABLE_INTERRUPT=0
run AMD_DIRECT_DISPATCH=0 HSA_ENABLE_INTERRUPT=0
run AMD_DIRECT_DISPATCH=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run AMD_DIRECT_DISPATCH=0 ROC_ACTIVE_WAIT_TIMEOUT=1000
run AMD_DIRECT_DISPATCH=0 DEBUG_CLR_MAX_BATCH_SIZE=1
run AMD_DIRECT_DISPATCH=0 DEBUG_CLR_MAX_BATCH_SIZE=16
run AMD_DIRECT_DISPATCH=0 GPU_FLUSH_ON_EXECUTION=1
run AMD_DIRECT_DISPATCH=0 HIP_FORCE_DEV_KERNARG=1
run AMD_DIRECT_DISPATCH=0 HIP_FORCE_DEV_KERNARG=0
run AMD_DIRECT_DISPATCH=0
