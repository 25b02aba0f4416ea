#!/bin/bash
# Run on the GPU box: bench stage times of the built library under every environment setting given as an argument
# ("VAMD_TONE_SPLIT=0" "VAMD_TONE_SPLIT=60" ...; test knobs are switched on), interleaved, twice, same box.
#   AB_WORKLOAD=c4 AB_STEPS=10 bash tools/ab_env.sh VAMD_TONE_SPLIT=0 VAMD_TONE_SPLIT=60
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for e in "$@"; do
    env VAMD_TEST_KNOBS=1 $e python bench.py --workload ${AB_WORKLOAD:-c4} --steps ${AB_STEPS:-10} --warmup 2 --no-cpu-baseline --no-parity-sample --no-neighbours --no-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$e', '%.2f M' % (d['value']/1e6), round(d['ms_per_step'],3), {a: round(b,3) for a,b in k.items()})"
  done
done
