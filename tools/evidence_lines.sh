R=${GRAFT_REPO_ROOT:-/root/repo}; E=$R/gpurun_out/evidence; mkdir -p $E; cd $R
python -c "import bench; print(bench.source_hash())" > $E/source_hash.txt
for w in c4 c5 c3 c2; do timeout 900 python bench.py --workload $w > $E/bench_$w.json 2> $E/bench_$w.err; done
: > $E/encode_loop.txt
for cfg in "1024 0.4 2646000" "4096 0.4 2646000" "65536 0.4 2646000" "1024 0.9 2646000" "65536 0.9 2646000"; do set -- $cfg
  LD_LIBRARY_PATH=build/dropin/ref:build/dropin ./build/dropin/encode_loop $1 $2 $3 write /tmp/el.pkts 2>> $E/encode_loop.txt
  LD_LIBRARY_PATH=build/dropin ./build/dropin/encode_loop $1 $2 $3 check /tmp/el.pkts 2>> $E/encode_loop.txt
done
cat $E/encode_loop.txt
