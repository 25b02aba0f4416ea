#!/bin/bash
# Run on the GPU box: N encoder threads through libvorbis' application loop, timed in C (oracle/ref_harness.c: ref_time_threads):
# the unmodified reference on the host CPUs, the hybrid libvorbis with a context per state, and through the batcher.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SECS=${SECS:-4}
echo "# host: $(nproc) hardware threads, cgroup cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for n in ${THREADS:-16 64 256}; do
  VAMD_CPU_ONLY=1 python tools/gpu_batcher_bench.py $n $SECS 2>&1 | tail -1
  for mb in ${BATCHES:-256}; do
    VAMD_BATCH=$mb python tools/gpu_batcher_bench.py $n $SECS 2>&1 | tail -3
  done
done
python tools/gpu_batcher_bench.py 16 $SECS 2>&1 | tail -1
