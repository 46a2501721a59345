#!/bin/bash
# One GPU-box session: run from the repo root as  bash tools/gpu_session.sh <tag> [steps...]
# steps: tests bench rehearse2 counters prof pmc mfma train codec   (default: tests bench)
set -o pipefail
: ${GRAFT_REPO_ROOT:=$(pwd)}      # the rocprofv3 steps cd to /tmp and name the repo absolutely
export GRAFT_REPO_ROOT
TAG=${1:-r02_x}; shift
STEPS=${@:-tests bench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for s in $STEPS; do
  echo "=== $s $(date +%T)"
  case $s in
    tests)     timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -ne 0 ] && exit $rc ;;
    bench)     timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; tail -c 600 $OUT/bench.json; [ $rc -ne 0 ] && { tail -20 $OUT/bench.err; exit $rc; } ;;
    benchq)    timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > $OUT/benchq.json 2> $OUT/benchq.err; rc=$?; tail -c 400 $OUT/benchq.json; [ $rc -ne 0 ] && { tail -20 $OUT/benchq.err; exit $rc; } ;;
    benchft)   FLOCODER_AMD_FUSED_TAIL=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_ft.json 2> $OUT/bench_ft.err; rc=$?; tail -c 300 $OUT/bench_ft.json; [ $rc -ne 0 ] && { tail -20 $OUT/bench_ft.err; exit $rc; } ;;
    stamps)    timeout -k 10 300 python tools/conv_stamps.py > $OUT/stamps.txt 2>&1 || { tail $OUT/stamps.txt; exit 1; }; tail -40 $OUT/stamps.txt ;;
    rehearse2) FLOCODER_AMD_SINGLE_GPU=1 FLOCODER_AMD_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-roofline > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; rc=$?; tail -c 400 $OUT/bench_2rank.json; [ $rc -ne 0 ] && { tail -20 $OUT/bench_2rank.err; exit $rc; } ;;
    counters)  rocprofv3 -L > $OUT/counters.txt 2>&1; grep -i -c mfma $OUT/counters.txt ;;
    optable)   timeout -k 10 300 python tools/op_table.py > $OUT/op_table.txt 2>&1 || exit 1; tail -3 $OUT/op_table.txt ;;
    prof)      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err) || { tail -20 $OUT/prof.err; exit 1; }
               find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \; ; head -12 $OUT/kernel_stats.csv ;;
    pmc)       (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_forward.py 3 > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1) &&
               (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/pmc_forward.py 3 > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1) || exit 1
               python tools/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv') $(find $OUT/pmc_write -name '*counter_collection.csv') 3 > $OUT/pmc_traffic.json; head -5 $OUT/pmc_traffic.json ;;
    pmccodec)  for what in sdvae_decode vqvae_encode vqvae_decode; do
                 (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_${what}_fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_codec.py $what 2 > $GRAFT_REPO_ROOT/$OUT/pmc_${what}_fetch.log 2>&1) &&
                 (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_${what}_write -- python3 $GRAFT_REPO_ROOT/tools/pmc_codec.py $what 2 > $GRAFT_REPO_ROOT/$OUT/pmc_${what}_write.log 2>&1) || { tail -5 $OUT/pmc_${what}_fetch.log $OUT/pmc_${what}_write.log; exit 1; }
                 python tools/pmc_summary.py $(find $OUT/pmc_${what}_fetch -name '*counter_collection.csv') $(find $OUT/pmc_${what}_write -name '*counter_collection.csv') 2 "tools/pmc_codec.py $what" > $OUT/pmc_traffic_${what}.json; head -4 $OUT/pmc_traffic_${what}.json
               done ;;
    mfma)      (cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/tools/pmc_forward.py 3 > $GRAFT_REPO_ROOT/$OUT/pmc_mfma.log 2>&1) || { tail -20 $OUT/pmc_mfma.log; exit 1; }
               python tools/pmc_mfma_summary.py $(find $OUT/pmc_mfma -name '*counter_collection.csv') > $OUT/pmc_mfma.json; head -30 $OUT/pmc_mfma.json ;;
    trainprof) (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trainprof -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 25 --warmup 5 > $GRAFT_REPO_ROOT/$OUT/trainprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/trainprof.err) || { tail -20 $OUT/trainprof.err; exit 1; }
               find $OUT/trainprof -name '*kernel_stats.csv' -exec cp {} $OUT/train_kernel_stats.csv \; ; python tools/train_profile_summary.py $OUT/trainprof 40 > $OUT/train_step_summary.txt; head -12 $OUT/train_step_summary.txt ;;
    trainprof32) (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trainprof32 -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 25 --warmup 5 --dim 32 --hw 32 --batch 64 --classes 102 > $GRAFT_REPO_ROOT/$OUT/trainprof32_bench.json 2> $GRAFT_REPO_ROOT/$OUT/trainprof32.err) || { tail -20 $OUT/trainprof32.err; exit 1; }
               find $OUT/trainprof32 -name '*kernel_stats.csv' -exec cp {} $OUT/train32_kernel_stats.csv \; ; python tools/train_profile_summary.py $OUT/trainprof32 40 > $OUT/train32_step_summary.txt; head -12 $OUT/train32_step_summary.txt ;;
    train32)   timeout -k 10 300 python tools/bench_train.py --dim 32 --hw 32 --batch 64 --classes 102 > $OUT/train32_bench.json 2> $OUT/train32.err || { tail -20 $OUT/train32.err; exit 1; }; cat $OUT/train32_bench.json ;;
    phases)    timeout -k 10 300 python tools/train_phases.py > $OUT/train_phases.txt 2>&1 && timeout -k 10 300 python tools/train_phases.py --dim 32 --hw 32 --batch 64 --classes 102 >> $OUT/train_phases.txt 2>&1; grep -v amdgpu.ids $OUT/train_phases.txt ;;
    train)     timeout -k 10 300 python tools/bench_train.py > $OUT/train_bench.json 2> $OUT/train.err || { tail -20 $OUT/train.err; exit 1; }; cat $OUT/train_bench.json ;;
    *)         echo "unknown step $s"; exit 2 ;;
  esac
done
echo "=== done $(date +%T)"
