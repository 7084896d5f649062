#!/bin/bash
# A/B of the bundle culling on the config-2 frame (tools/rank_times.py: kernel time per rank at world 1/2/4/8).
cd $GRAFT_REPO_ROOT
for c in 1 0; do
  echo "== RTK_BUNDLE_CULL=$c"
  RTK_BUNDLE_CULL=$c TC_MODES="${TC_MODES:-2 3 4}" python tools/rank_times.py 2>&1 | grep -E "^world"
done
