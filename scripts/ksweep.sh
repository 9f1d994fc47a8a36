#!/bin/bash
# development aid: PG maxupd=1 kernel time on the C2 shape for several k (line-granularity of the gather)
for k in "$@"; do
  python bench.py --steps 10 --warmup 3 --maxupd 1 --k $k --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('k=$k', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms_per_sweep'],4), 'frac', round(d['roofline']['frac'],3))"
done
