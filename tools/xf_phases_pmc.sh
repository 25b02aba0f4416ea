#!/bin/bash
# Run on the GPU box: k_transform<11> phase by phase.  ab/libX<k>.so are scratch builds (a patch that wraps phase k of transform_block in
# `if (VAMD_XF_SKIP != k)`; not in the tree) with -DVAMD_XF_SKIP=k (phase k left out; k = 0: the
# full kernel); per build one counter pass over 32 768 stereo blocks and the stage's HIP-event time.  A phase's share = full - skipped.
#   1 window  2 fold  3 butterfly stages  4 levels 32/16/8  5 bit-reverse + rotate  6 spectrum out  7 FFT ido 1+4  8 FFT ido 16
# PFX=P tools/xf_phases_pmc.sh 0 1 compares two whole builds ab/libP0.so, ab/libP1.so the same way.
#   9 FFT ido 64  10 FFT last radix-4 + radix-2  11 logfft + run peaks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cp $R/vorbis_amd/libvorbis_amd.so /tmp/keep.so
PFX=${PFX:-X}
for k in ${@:-0 1 2 3 4 5 6 7 8 9 10 11}; do
  cp $R/ab/lib$PFX$k.so $R/vorbis_amd/libvorbis_amd.so
  rm -rf /tmp/p
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES -d /tmp/p -o x -- python $R/tools/prof_run.py 32768 1 > /dev/null 2> /tmp/p.log
  rm -rf /tmp/q
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY -d /tmp/q -o x -- python $R/tools/prof_run.py 32768 1 > /dev/null 2> /tmp/q.log
  ms=$(python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-sample --no-neighbours 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['roofline']['kernels_ms_per_step']['transform'],3))")
  echo "skip $k  transform ${ms} ms  $(python $R/tools/pmc_summary.py /tmp/p/x_results.db | grep 'k_transform')  $(python $R/tools/pmc_summary.py /tmp/q/x_results.db | grep 'k_transform' | sed 's/.*waves [0-9]* //')"
done
cp /tmp/keep.so $R/vorbis_amd/libvorbis_amd.so
