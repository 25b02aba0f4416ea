#!/bin/bash
# Run on the GPU box: the bench line's value a few times per workload in separate processes (run-to-run spread on one box).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import bench; print('# source_hash', bench.source_hash())"
for w in c4 c5 c3 c2; do
  for rep in 1 2 3; do
    python bench.py --workload $w --no-cpu-baseline --no-neighbours --no-workloads --no-parity-sample 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', 'value', round(d['value']), d['unit'], ' ms_per_step', round(d['ms_per_step'],3), ' frac', round(d['roofline']['frac'],4))"
  done
done
