#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: per-wave instruction counters of one kernel for every library variant ab/lib*.so.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cp $R/vorbis_amd/libvorbis_amd.so /tmp/keep.so
for v in $R/ab/lib*.so; do
  cp $v $R/vorbis_amd/libvorbis_amd.so
  rm -rf /tmp/p
  VAMD_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc ${2:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES} -d /tmp/p -o x -- python $R/tools/prof_run.py 16384 1 > /dev/null 2> /tmp/p.log
  echo "== $(basename $v)"
  python $R/tools/pmc_summary.py /tmp/p/x_results.db | grep "${1:-k_floor}"
done
cp /tmp/keep.so $R/vorbis_amd/libvorbis_amd.so
