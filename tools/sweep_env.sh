#!/bin/bash
# Config-2 frame time (tools/rank_times.py, world 1) under a few environment knobs.  usage: tools/sweep_env.sh  (on the GPU box)
run() { echo -n "$* : "; env "$@" TC_MODES=3 python tools/rank_times.py 2>&1 | grep "world 1" | sed 's/.*slowest rank \([0-9.]* ms\).*/\1/'; }
run RTK_SLICE_MIN_TRIS=12
run RTK_SLICE_MIN_TRIS=8
run RTK_SLICE_MIN_TRIS=16
run RTK_SLICE_MIN_TRIS=24
run RTK_COST_RESORT_EVERY=32
run RTK_COST_RESORT_EVERY=2
run RTK_LIGHT_BELOW_CYCLES=20000
run RTK_LIGHT_BELOW_CYCLES=100000
run RTK_SHADOW_EARLY_EXIT=0
run RTK_COST_FEEDBACK=0
