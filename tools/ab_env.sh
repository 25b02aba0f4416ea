#!/bin/bash
# Run on the GPU box: the detector kernels' durations in the C5 workload for every library variant ab/lib*.so.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cp $R/vorbis_amd/libvorbis_amd.so /tmp/keep.so
for v in $R/ab/lib*.so; do
  cp $v $R/vorbis_amd/libvorbis_amd.so
  rm -rf /tmp/k5
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k5 -o kt -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-sample > /dev/null 2>&1
  echo "== $(basename $v)"
  python $R/tools/kt_summary.py /tmp/k5/kt_results.db | grep "${1:-k_env}"
done
cp /tmp/keep.so $R/vorbis_amd/libvorbis_amd.so
