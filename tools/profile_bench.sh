#!/bin/bash
# Round profile of the benchmark command: kernel-trace stats + PMC passes (each in its own run, as the guide prescribes).
# usage: tools/profile_bench.sh <tag> [bench args...]   (run on the GPU box; outputs under gpurun_out/<tag>/)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 50 --warmup 5 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 5 --warmup 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B --steps 5 --warmup 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $B --steps 5 --warmup 2 > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq2 -- $B --steps 5 --warmup 2 > $OUT/pmc_sq2.log 2>&1
# the slow workloads of the bench line's `extras` (configs 3 and 4's shape): per-kernel times of the streaming pipeline
TC_MODES="0" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_extras -- python3 $GRAFT_REPO_ROOT/tools/time_configs.py cfg3 cfg4 > $OUT/trace_extras.log 2>&1
cat $OUT/trace/*/*kernel_stats.csv | cut -c1-160
