#!/bin/bash
# Run on the GPU box: counters of the residue search and the packet assembly of a batch (k_residue_chunks, k_pack_waves) --
# instruction counts, lane utilisation and HBM bytes per stereo long block, each counter set in a pass of its own
# (never with a trace domain) over tools/res_profile.py's batch of 65 536 blocks (8 analyze calls of it per run).
R=${GRAFT_REPO_ROOT:-/root/repo}
NB=${1:-65536}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pr
  timeout 300 rocprofv3 --pmc $set -d /tmp/pr -o x -- python $R/tools/res_profile.py 44k_stereo_q4 $NB > /dev/null 2> /tmp/pr.log
  echo "== rocprofv3 --pmc $set -- python tools/res_profile.py 44k_stereo_q4 $NB"
  python $R/tools/pmc_summary.py /tmp/pr/x_results.db | grep -i "k_residue\|k_pack\|k_couple\|k_calib\|^kernel"
done
