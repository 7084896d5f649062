#!/bin/bash
# The streaming pipeline's knobs on the config 3 / 4 / 5 shapes.  usage: tools/sweep_stream.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" TC_MODES=0 python tools/time_configs.py cfg3 cfg4 cfg5 2>&1 | grep "cfg" | grep -v "spp1" | awk '{print "   ", $1, $2, $3, $4, $5, $6, $7, $8}'; }
run RTK_STREAM_DEEP_LEVEL=99
run RTK_STREAM_DEEP_LEVEL=2 RTK_STREAM_DEEP_MODE=0
run RTK_STREAM_DEEP_LEVEL=3 RTK_STREAM_DEEP_MODE=0
run RTK_STREAM_DEEP_LEVEL=2 RTK_STREAM_DEEP_MODE=2
run RTK_STREAM_SORT_FROM=2
