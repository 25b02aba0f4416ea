#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: bytes k_tone_chase fetches per stereo block, with the tone chain beside k_noise and after it
# (FETCH_SIZE in the profile's units: x2 x 32 B per count on gfx950, tools/make_profiles.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ov in "" "VAMD_NO_OVERLAP=1"; do
  rm -rf /tmp/p
  env $ov timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p -o x -- python $R/tools/prof_run.py 131072 1 > /dev/null 2> /tmp/p.log
  echo "== ${ov:-overlap}"
  python $R/tools/prof_summary.py pmc /tmp/p/x_results.db 2>&1 | grep -i "chase\|floor\|copy\|elementwise" | head -6
done
