#!/bin/bash
# Run on the GPU box: the floor stage with one channel-block per wave (k_floor) against the two channels of a stereo block
# per wave (k_floor_pair) at several occupancies (ab/libP<waves per SIMD>.so, tools/build_variant.sh), interleaved on one
# box: C4 (long blocks, q 0.4) and C5 (mixed, q 0.9) with the pairing off, on for short blocks only, for long only.
export VAMD_TEST_KNOBS=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp vorbis_amd/libvorbis_amd.so /tmp/keep.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$1', round(d['ms_per_step'],3), round(d['value']/1e6,2), {a: round(b,3) for a,b in k.items() if a in ('floor','couple','noisemask')})"; }
for v in ab/libP*.so; do
  cp $v vorbis_amd/libvorbis_amd.so
  for mode in "off 2000000000 3" "short 0 1" "long 0 2"; do
    set -- $mode
    [ "$1" = short ] || VAMD_FLOOR_PAIR_MIN=$2 VAMD_FLOOR_PAIR_W=$3 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-sample --no-neighbours --no-workloads 2>/dev/null | line "$(basename $v) c4 pair=$1"
    [ "$1" = long ] || VAMD_FLOOR_PAIR_MIN=$2 VAMD_FLOOR_PAIR_W=$3 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-sample 2>/dev/null | line "$(basename $v) c5 pair=$1"
  done
done
cp /tmp/keep.so vorbis_amd/libvorbis_amd.so
