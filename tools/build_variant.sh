#!/bin/bash
# Build a scratch variant of the library into ab/lib<name>.so (git-ignored; travels to the GPU box with gpurun):
#   tools/build_variant.sh <name> [-DFLAG=...]...      then on the box: bash tools/ab.sh
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -Ivorbis_amd/csrc "$@" \
  vorbis_amd/csrc/vamd_hip.hip vorbis_amd/csrc/vamd_batcher.hip vorbis_amd/csrc/vamd_feed.hip -o ab/lib$name.so 2>/dev/null && ls -la ab/lib$name.so
