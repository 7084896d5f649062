#!/bin/bash
# bench.py headline (no CPU baseline, no extras) under values of an environment knob.  usage: tools/ab_env_bench.sh RTK_KNOB v1 v2 ...  (on the GPU box)
cd $GRAFT_REPO_ROOT
K=$1; shift
for v in "$@" "$@"; do
  env $K=$v python bench.py --no-cpu-baseline --no-extras 2>/dev/null > /tmp/ab_bench.json
  python -c "import json; d=json.load(open('/tmp/ab_bench.json')); print('$K=$v', 'ms_per_step', round(d['ms_per_step'],4), 'Mrays/s', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'critical', round(d['critical_path_ms'],4), 'verified', d['verified'])"
done
