#!/bin/bash
# Sweep of the streaming pipeline's samples-per-launch (RTK_STREAM_BATCH) and batches in flight (RTK_STREAM_LANES) on the
# BASELINE config shapes.  usage (GPU box): tools/sweep_batch.sh "cfg3 spp4" "1:4 4:1 2:2" ...
CASE=$1; shift
for bl in $*; do
  b=${bl%%:*}; l=${bl##*:}
  echo "== batch $b lanes $l"
  RTK_STREAM_BATCH=$b RTK_STREAM_LANES=$l TC_MODES=6 TC_WARM=${TC_WARM:-2} TC_REPS=${TC_REPS:-3} python3 tools/time_configs.py "$CASE" 2>&1 | tail -2
done
