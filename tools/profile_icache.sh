#!/bin/bash
# Run on the GPU box: instruction-cache counters per kernel (own pass).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_icache.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  rm -rf /tmp/p
  timeout 300 rocprofv3 --pmc $set -d /tmp/p -o x -- python $R/tools/prof_run.py 65536 1 > /dev/null 2> /tmp/p.log
  echo "== rocprofv3 --pmc $set -- python tools/prof_run.py 65536 1" >> $O/pmc_icache.txt
  python $R/tools/pmc_summary.py /tmp/p/x_results.db >> $O/pmc_icache.txt 2>&1
  tail -2 /tmp/p.log >> $O/pmc_icache.txt
done
