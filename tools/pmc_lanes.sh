#!/bin/bash
# Run on the GPU box: lane utilisation of the vector instructions, per kernel: of the 64 lanes a VALU
# instruction could drive, how many did.  SQ_THREAD_CYCLES_VALU counts lane-cycles with the lane's exec bit set,
# SQ_ACTIVE_INST_VALU the cycles the vector unit spent on instructions (four per wave64 instruction): their ratio / 16 is the
# mean fraction of lanes enabled.  One counter pass over one step of 131 072 stereo blocks (tools/prof_run.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
NB=${1:-131072}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_THREAD_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\|SQ_INSTS_VALU\b" | sort | uniq -c
rm -rf /tmp/pl
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d /tmp/pl -o x -- python $R/tools/prof_run.py $NB 1 > /dev/null 2> /tmp/pl.log
echo "== rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -- python tools/prof_run.py $NB 1"
python $R/tools/pmc_summary.py /tmp/pl/x_results.db
tail -3 /tmp/pl.log
