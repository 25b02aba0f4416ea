#!/bin/bash
# Run on the GPU box: parity spot-check, then the bench's per-stage times for a list of environment settings, interleaved.
#   tools/ab_env2.sh "VAMD_NOISE_GANG=1" "VAMD_NOISE_GANG=2"
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for v in "$@"; do
    env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-sample --no-neighbours --no-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$v', round(d['ms_per_step'],3), {a: round(b,3) for a,b in k.items()})"
  done
done
