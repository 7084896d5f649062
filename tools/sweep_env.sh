#!/bin/bash
# Config-2 frame time (tools/rank_times.py, world 1 only) under environment knobs.  usage: tools/sweep_env.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" TC_WORLDS=1 TC_MODES=${TC_MODES:-0} python tools/rank_times.py 2>&1 | grep "world 1" | sed 's/.*slowest rank \([0-9.]* ms\).*/\1/' | tr '\n' ' '; echo; }
for light in ${LIGHTS:-20000 40000 80000 160000 400000}; do
  for smin in ${SMINS:-65}; do
    run RTK_LIGHT_BELOW_CYCLES=$light RTK_SLICE_MIN_TRIS=$smin
  done
done
