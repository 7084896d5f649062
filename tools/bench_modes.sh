#!/bin/bash
# quick A/B of the three traversal strategies on the benchmark workload (no CPU baseline)
for m in ${MODES:-1 2 3 6}; do
  python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --trace-mode $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['trace_mode'], round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms  kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
done
