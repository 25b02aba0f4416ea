cd $GRAFT_REPO_ROOT
for L in H N H N; do
cp ab/lib$L.so vorbis_amd/libvorbis_amd.so
echo "== lib $L"; python tools/gpu_block_latency.py 2>/dev/null | head -1
done
python tools/gpu_block_phases.py 2>&1 | tail -5
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
