#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: each kernel's own duration (tone chain serialised) for every library variant ab/lib*.so.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cp $R/vorbis_amd/libvorbis_amd.so /tmp/keep.so
for v in $R/ab/lib*.so; do
  cp $v $R/vorbis_amd/libvorbis_amd.so
  rm -rf /tmp/kts
  VAMD_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python $R/tools/prof_run.py ${1:-131072} 3 > /dev/null 2> /tmp/kts.log
  echo "== $(basename $v)"
  python $R/tools/kt_summary.py /tmp/kts/kt_results.db | grep -v "mdct_only"
done
cp /tmp/keep.so $R/vorbis_amd/libvorbis_amd.so
