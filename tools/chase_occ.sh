#!/bin/bash
# Run on the GPU box: k_tone_chase held to fewer waves per CU (LDS padding) -- what it fetches (its live lines then fit
# in L2 and the second half of a 128-byte line is a hit) against what the step costs.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for pad in 0 2048 5632 12288; do
  rm -rf /tmp/p
  env VAMD_CHASE_LDS_PAD=$pad timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p -o x -- python $R/tools/prof_run.py 131072 1 > /dev/null 2> /tmp/p.log
  echo "== pad $pad: $(python $R/tools/prof_summary.py pmc /tmp/p/x_results.db 2>&1 | grep -i "chase" | head -1)"
  cd $R; env VAMD_CHASE_LDS_PAD=$pad python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],3), {a: round(b,3) for a,b in k.items()})"; cd /tmp
done
