#!/bin/bash
# ISA of one kernel out of the product's device code: tools/kernel_isa.sh <mangled-name-prefix> [extra flags] > file.s
# (e.g. _Z7k_noiseILi10ELi2EE); with no argument lists the kernel symbols.
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Ivorbis_amd/csrc "$@" \
  --cuda-device-only -S vorbis_amd/csrc/vamd_hip.hip -o /tmp/vamd_all.s 2>/dev/null
if [ -z "$name" ]; then grep -o "^_Z[A-Za-z0-9_]*:" /tmp/vamd_all.s | grep "k_" ; exit; fi
awk -v n="$name" 'index($0, n) == 1 && /:/ {p=1} p{print} p && /s_endpgm/{exit}' /tmp/vamd_all.s
