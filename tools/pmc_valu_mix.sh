#!/bin/bash
# Run on the GPU box: what KIND of vector instructions each kernel of one full-analysis step issues (rocprofv3 --pmc, one set per
# pass) -- the dynamic mix behind roofline.valu: tools/micro/chip_rate.hip prices an add / mul / mov-class instruction at ~2.7 SIMD
# cycles, everything in the three-operand encodings, conversions, compares, shifts and fp64 at ~4.7, transcendentals at 8.7.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/p
  timeout 300 rocprofv3 --pmc $set -d /tmp/p -o x -- python $R/tools/prof_run.py ${1:-131072} 1 > /dev/null 2> /tmp/p.log
  echo "== rocprofv3 --pmc $set -- python tools/prof_run.py ${1:-131072} 1"
  python $R/tools/pmc_summary.py /tmp/p/x_results.db | grep "${2:-k_}"
done
