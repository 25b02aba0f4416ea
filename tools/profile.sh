#!/bin/bash
# Run on the GPU box (through gpurun): collects the rocprofv3 evidence for profiles/.
# Counters go in their own passes (never combined with trace domains other than kernel-trace).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-neighbours > $O/bench_under_rocprof.json 2> $O/kt_bench.log
python $R/tools/prof_summary.py kt $O/kt_bench/kt_results.db > $O/kernel_trace_stats.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/tools/prof_run.py 65536 1 > /dev/null 2> $O/pmc_fetch.log
python $R/tools/prof_summary.py pmc $O/pmc_fetch/f_results.db > $O/pmc_fetch_size.txt 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/tools/prof_run.py 65536 1 > /dev/null 2> $O/pmc_write.log
python $R/tools/prof_summary.py pmc $O/pmc_write/w_results.db > $O/pmc_write_size.txt 2>&1
# BASELINE config 5 (mixed short / long streams): kernel trace of the same bench workload
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_c5 -o kt -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_under_rocprof.json 2> $O/kt_c5.log
python $R/tools/prof_summary.py kt $O/kt_c5/kt_results.db > $O/kernel_trace_stats_c5.txt 2>&1
rm -rf $O/kt_bench $O/pmc_fetch $O/pmc_write $O/kt_c5
ls -la $O
