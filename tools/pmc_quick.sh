#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: per-wave SQ counters of every kernel for one full-analysis step (serial tone chain).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVES"; do
  rm -rf /tmp/p
  VAMD_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc $set -d /tmp/p -o x -- python $R/tools/prof_run.py ${1:-32768} 1 > /dev/null 2> /tmp/p.log
  echo "== $set"
  python $R/tools/pmc_summary.py /tmp/p/x_results.db | grep "${2:-k_}"
done
