#!/bin/bash
# Run on the GPU box: k_floor's instructions per wave when the kernel ends after phase k (ab/libS<k>.so, built with
# -DVAMD_STOP_AFTER=k), i.e. the running total of dynamic instructions phase by phase.
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/vorbis_amd/libvorbis_amd.so /tmp/keep.so
for v in $R/ab/libS*.so; do
  cp $v $R/vorbis_amd/libvorbis_amd.so
  echo "== $(basename $v)"
  bash $R/tools/pmc_set.sh "${1:-k_floor}" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
done
cp /tmp/keep.so $R/vorbis_amd/libvorbis_amd.so
