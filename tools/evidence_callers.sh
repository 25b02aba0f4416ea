# Run on the GPU box: the callers' figures that depend on the BINDING only (integration/*.c is outside the kernels' source hash):
# the application loop on the drop-in library, and one thread with large writes through the hybrid.
R=${GRAFT_REPO_ROOT:-/root/repo}; E=$R/gpurun_out/evidence; mkdir -p $E; cd $R
: > $E/encode_loop.txt
for cfg in "1024 0.4 2646000" "4096 0.4 2646000" "65536 0.4 2646000" "1024 0.9 2646000" "65536 0.9 2646000"; do set -- $cfg
  LD_LIBRARY_PATH=build/dropin/ref:build/dropin ./build/dropin/encode_loop $1 $2 $3 write /tmp/el.pkts 2>> $E/encode_loop.txt
  LD_LIBRARY_PATH=build/dropin ./build/dropin/encode_loop $1 $2 $3 check /tmp/el.pkts 2>> $E/encode_loop.txt
done
echo "# VAMD_DETECTOR=gpu (the GPU's detector for every stream, as rounds 2-5 ran):" >> $E/encode_loop.txt
for cfg in "1024 0.4 2646000" "1024 0.9 2646000"; do set -- $cfg
  LD_LIBRARY_PATH=build/dropin/ref:build/dropin ./build/dropin/encode_loop $1 $2 $3 write /tmp/el.pkts 2> /dev/null
  VAMD_DETECTOR=gpu LD_LIBRARY_PATH=build/dropin ./build/dropin/encode_loop $1 $2 $3 check /tmp/el.pkts 2>> $E/encode_loop.txt
done
: > $E/lookahead.txt
for k in "60 0.4 s16" "60 0.9 gated" "30 0.4 s16 128000"; do timeout 400 python tools/gpu_lookahead_bench.py $k >> $E/lookahead.txt 2>/dev/null; done
cat $E/encode_loop.txt | cut -c1-200; cat $E/lookahead.txt
