#!/bin/bash
# sweeps the streaming pipeline's deep-level strategy on one config (arg: substring of the case name)
for lvl in 1 2 3 99; do for mode in 0 1; do for ml in 12 32; do
  [ $lvl = 99 ] && [ "$mode$ml" != "012" ] && continue
  [ $mode = 1 ] && [ $ml = 32 ] && continue
  echo -n "deep_level=$lvl mode=$mode min_lanes=$ml: "
  RTK_STREAM_DEEP_LEVEL=$lvl RTK_STREAM_DEEP_MODE=$mode RTK_AUTO_MIN_LANES=$ml python tools/time_configs.py "$1" 2>/dev/null | grep " stream "
done; done; done
