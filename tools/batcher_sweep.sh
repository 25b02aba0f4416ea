#!/bin/bash
# Run on the GPU box: N encoder threads through libvorbis' application loop, timed in C (oracle/ref_harness.c: ref_time_threads):
# the unmodified reference on the host CPUs, and the hybrid libvorbis through the batcher; then a context per state.
# Streams are long enough that every row runs for seconds (the host's CPU quota is enforced per 100 ms period: a run
# shorter than that borrows from the next period and reads several times too fast).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "# host: $(nproc) hardware threads, cgroup cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for n in ${THREADS:-16 64 256}; do
  secs=$(( 7680 / n )); [ $secs -gt 120 ] && secs=120; [ $secs -lt 20 ] && secs=20
  VAMD_CPU_ONLY=1 python tools/gpu_batcher_bench.py $n $secs 2>&1 | tail -1
  for mb in ${BATCHES:-256}; do
    VAMD_BATCH=$mb python tools/gpu_batcher_bench.py $n $secs 2>&1 | tail -3
  done
done
[ -n "${SKIP_PER_STATE:-}" ] || python tools/gpu_batcher_bench.py 16 20 2>&1 | tail -1
