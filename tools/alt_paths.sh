#!/bin/bash
export VAMD_TEST_KNOBS=1  # the knobs below are test knobs: ignored without this (vorbis_amd/csrc/vamd_knobs.h)
# Run on the GPU box: the GPU suite once more with every batch-size-dependent choice of kernel turned the other way (the
# environment of INTEGRATION.md's last section), so that the kernels a default run of the suite reaches only at sizes it
# does not use are held to the same oracle.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import bench; print('# source_hash', bench.source_hash())"
run() { echo "== $*"; env "$@" timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2; }
run VAMD_MASKS_SEPARATE=1 VAMD_PACK_PAIR_MAX=0 VAMD_RES_TEAM_MAX=0
run VAMD_CHASE_WAVE_MAX=0 VAMD_NO_OVERLAP=1
run VAMD_STAGE_COPIES=1 VAMD_FOLD_SEPARATE=1
run VAMD_FLOOR_PAIR_MIN=0                    # two channels per wave in the floor stage at every size, both size classes
run VAMD_FLOOR_PAIR_MIN=2000000000           # ... and never
run VAMD_RES_IN_LDS=1 VAMD_PACK_PER_PACKET=1 VAMD_NOISE_WAVES=28   # round 6's batch kernels (k_residue_chunks, k_pack_waves) off; seven noise teams
