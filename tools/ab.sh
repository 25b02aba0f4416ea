#!/bin/bash
# Run on the GPU box: the bench's per-stage times for library variants ab/lib*.so, interleaved, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp vorbis_amd/libvorbis_amd.so /tmp/keep.so
for rep in 1 2; do
  for v in ab/lib*.so; do
    cp $v vorbis_amd/libvorbis_amd.so
    python bench.py --workload ${AB_WORKLOAD:-c4} --steps ${1:-10} --warmup 2 --no-cpu-baseline --no-parity-sample --no-neighbours --no-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$v', '%.2f M' % (d['value']/1e6), round(d['ms_per_step'],3), {a: round(b,3) for a,b in k.items()})"
  done
done
cp /tmp/keep.so vorbis_amd/libvorbis_amd.so
