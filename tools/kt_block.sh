#!/bin/bash
# Run on the GPU box: per-kernel durations of the per-block entry point (one stereo block per call).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktb
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktb -o kt -- python $R/tools/gpu_block_latency.py > /dev/null 2> /tmp/ktb.log
python $R/tools/kt_summary.py /tmp/ktb/kt_results.db
python $R/tools/kt_timeline.py /tmp/ktb/kt_results.db k_transform
