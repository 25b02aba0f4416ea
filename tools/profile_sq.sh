#!/bin/bash
# Run on the GPU box: instruction-mix / LDS counters per kernel (each set in its own pass), per wave.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_sq_counters.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/p
  timeout 300 rocprofv3 --pmc $set -d /tmp/p -o x -- python $R/tools/prof_run.py 65536 1 > /dev/null 2> /tmp/p.log
  echo "== rocprofv3 --pmc $set -- python tools/prof_run.py 65536 1" >> $O/pmc_sq_counters.txt
  python $R/tools/pmc_summary.py /tmp/p/x_results.db >> $O/pmc_sq_counters.txt
done
