#!/bin/bash
# Run on the GPU box (through gpurun): collects the rocprofv3 evidence for profiles/ (tools/make_profiles.py turns it
# into the committed summaries).  Counters go in their own passes, never combined with a trace domain.
#   C4 (the bench default): kernel trace of bench.py; FETCH_SIZE / WRITE_SIZE / four SQ sets over 131 072 stereo blocks
#   C5 (mixed short / long streams): kernel trace of bench.py --workload c5; the same counter passes over one c5 step
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profile
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NB=${1:-131072}
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-neighbours --no-workloads > $O/bench_under_rocprof.json 2> $O/kt_bench.log
python $R/tools/prof_summary.py kt $O/kt_bench/kt_results.db > $O/kernel_trace_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_c5 -o kt -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_under_rocprof.json 2> $O/kt_c5.log
python $R/tools/prof_summary.py kt $O/kt_c5/kt_results.db > $O/kernel_trace_stats_c5.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr -d $O/pmc_$ctr -o x -- python $R/tools/prof_run.py $NB 1 > /dev/null 2> $O/pmc_$ctr.log
  python $R/tools/prof_summary.py pmc $O/pmc_$ctr/x_results.db > $O/pmc_$ctr.txt 2>&1
  timeout 600 rocprofv3 --pmc $ctr -d $O/pmc_c5_$ctr -o x -- python $R/tools/prof_run_c5.py 1 > $O/c5_step.json 2> $O/pmc_c5_$ctr.log
  python $R/tools/prof_summary.py pmc $O/pmc_c5_$ctr/x_results.db > $O/pmc_c5_$ctr.txt 2>&1
done
: > $O/pmc_sq_counters.txt
: > $O/pmc_sq_counters_c5.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32"; do
  rm -rf /tmp/p
  timeout 300 rocprofv3 --pmc $set -d /tmp/p -o x -- python $R/tools/prof_run.py $NB 1 > /dev/null 2> /tmp/p.log
  echo "== rocprofv3 --pmc $set -- python tools/prof_run.py $NB 1" >> $O/pmc_sq_counters.txt
  python $R/tools/pmc_summary.py /tmp/p/x_results.db >> $O/pmc_sq_counters.txt
  rm -rf /tmp/p
  timeout 300 rocprofv3 --pmc $set -d /tmp/p -o x -- python $R/tools/prof_run_c5.py 1 > /dev/null 2> /tmp/p.log
  echo "== rocprofv3 --pmc $set -- python tools/prof_run_c5.py 1" >> $O/pmc_sq_counters_c5.txt
  python $R/tools/pmc_summary.py /tmp/p/x_results.db >> $O/pmc_sq_counters_c5.txt
done
bash $R/tools/pmc_valu_mix.sh $NB > $O/pmc_valu_mix.txt 2>&1   # the dynamic instruction mix behind roofline.valu
rm -rf $O/kt_bench $O/kt_c5 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_c5_FETCH_SIZE $O/pmc_c5_WRITE_SIZE
ls -la $O
