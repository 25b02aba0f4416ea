#!/bin/bash
# Run on the GPU box: the bench's step and stage times with an environment variable at several values, interleaved.
#   tools/env_ab.sh NAME v1 v2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
name=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $name=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-sample --no-neighbours --no-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('$name=$v', round(d['ms_per_step'],3), {a: round(b,3) for a,b in k.items()})"
  done
done
