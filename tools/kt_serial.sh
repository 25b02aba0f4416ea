#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: per-kernel durations with the tone chain serialised behind k_noise (no co-residency),
# which is what each kernel costs on its own.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/profile
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kts
VAMD_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python $R/tools/prof_run.py ${1:-131072} 3 > /dev/null 2> /tmp/kts.log
python $R/tools/prof_summary.py kt /tmp/kts/kt_results.db > $R/gpurun_out/profile/kernel_trace_serial.txt 2>&1
cat $R/gpurun_out/profile/kernel_trace_serial.txt
