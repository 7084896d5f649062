#!/bin/bash
# Round profile (run on the GPU box): the benchmark command and the `extras` workloads, each under rocprofv3 --kernel-trace --stats
# and under separate --pmc passes (MI355X_MICROARCH.md: counters in their own runs).  Outputs under gpurun_out/<tag>/.
# usage: tools/profile_round.sh <tag> [workload ...]      default workloads: bench config3 config4 config5 synthetic_*
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
WL=${@:-bench config3 config4 config5 synthetic_coherent_primary synthetic_shuffled_primary synthetic_uniform_secondary}
cd /tmp && export TMPDIR=/tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES")
for w in $WL; do
  if [ "$w" = bench ]; then
    CMD="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-verify"
    TR="$CMD --steps 50 --warmup 5"; PM="$CMD --steps 5 --warmup 20"
  else
    TR="python3 $GRAFT_REPO_ROOT/tools/profile_workloads.py $w 4"; PM="$TR"
  fi
  mkdir -p $OUT/$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace -- $TR > $OUT/$w/trace.log 2>&1
  echo "[$w] trace done" 
  i=0
  for P in "${PASSES[@]}"; do
    rocprofv3 --pmc $P --output-format csv -d $OUT/$w/pmc_$i -- $PM > $OUT/$w/pmc_$i.log 2>&1
    i=$((i+1))
  done
  echo "[$w] pmc done"
done
(cd $GRAFT_REPO_ROOT && git rev-parse HEAD 2>/dev/null || echo unknown) > $OUT/git_head.txt
