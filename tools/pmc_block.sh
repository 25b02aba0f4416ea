#!/bin/bash
# Run on the GPU box: SQ counters per wave for the kernels of the per-block entry point (one stereo block per call):
# instructions, memory reads, and how much of a lone wave's life is waiting (SQ_WAIT_ANY against SQ_WAVE_CYCLES).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
rm -rf /tmp/p
timeout 300 rocprofv3 --pmc $set -d /tmp/p -o x -- python $R/tools/gpu_block_latency.py > /dev/null 2> /tmp/p.log
python $R/tools/pmc_summary.py /tmp/p/x_results.db | grep "k_pack\|k_floor\|k_tone_chase_wave\|k_tone_seed\|k_noise\|k_transform\|k_couple\|k_residue"
done
