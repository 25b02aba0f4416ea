#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: k_floor's time against blocks in flight per CU (LDS padding limits residency).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for pad in 0 4000 6000 9600; do
  VAMD_FLOOR_LDS_PAD=$pad python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-sample --no-neighbours --no-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad', $pad, d['roofline']['kernels_ms_per_step']['floor'])"
done
