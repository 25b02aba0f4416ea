#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: one set of SQ counters per wave for the kernels matching a pattern.
#   tools/pmc_set.sh "<kernel grep pattern>" COUNTER [COUNTER ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p
VAMD_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc "$@" -d /tmp/p -o x -- python $R/tools/prof_run.py 32768 1 > /dev/null 2> /tmp/p.log
python $R/tools/pmc_summary.py /tmp/p/x_results.db | grep "$pat" || tail -5 /tmp/p.log
