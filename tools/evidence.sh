#!/bin/bash
# Run on the GPU box (one gpurun call): everything a round's committed evidence is made from, on ONE build --
# the GPU test suite, the soak, the counter / trace passes (tools/profile.sh), the bench lines, the batcher sweep and the
# per-block path.  Outputs under gpurun_out/ (evidence/, profile/, soak.txt); tools/make_profiles.py and a few copies
# turn them into profiles/rNN_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/evidence
rm -rf $E; mkdir -p $E
cd $R
python -c "import bench; print(bench.source_hash())" > $E/source_hash.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $E/pytest_gpu.txt
timeout 900 python tools/soak.py ${SOAK:-20000} $R/gpurun_out/soak.txt > /dev/null 2>&1
tail -3 $R/gpurun_out/soak.txt > $E/soak_tail.txt
bash tools/profile.sh > $E/profile.log 2>&1
for w in c4 c5 c3 c2; do
  timeout 900 python bench.py --workload $w > $E/bench_$w.json 2> $E/bench_$w.err
done
timeout 900 bash tools/batcher_sweep.sh > $E/batcher.txt 2>&1
timeout 300 bash tools/kt_block.sh > $E/block_path.txt 2>&1
timeout 300 python tools/gpu_block_latency.py >> $E/block_path.txt 2>/dev/null
timeout 300 python tools/gpu_block_phases.py >> $E/block_path.txt 2>/dev/null
: > $E/lookahead.txt
for k in "60 0.4 s16" "60 0.9 gated" "30 0.4 s16 128000"; do timeout 400 python tools/gpu_lookahead_bench.py $k >> $E/lookahead.txt 2>/dev/null; done
timeout 600 python tools/soak_lookahead.py 60 $E/soak_lookahead.txt > /dev/null 2>&1
timeout 900 bash tools/alt_paths.sh > $E/alt_paths.txt 2>&1
# the residue search and the packet assembly of a batch (round 6: k_residue_chunks, k_pack_waves): stage times + phase stopwatch,
# then their counters; the detector's spectrum kernel phase by phase
{ timeout 300 python tools/res_profile.py 44k_stereo_q4 65536; echo "---- the same through the work vector in LDS and a workgroup per packet (VAMD_RES_IN_LDS, VAMD_PACK_PER_PACKET)";
  VAMD_TEST_KNOBS=1 VAMD_RES_IN_LDS=1 VAMD_PACK_PER_PACKET=1 timeout 300 python tools/res_profile.py 44k_stereo_q4 65536; } > $E/res_pack.txt 2>/dev/null
timeout 900 bash tools/pmc_res.sh >> $E/res_pack.txt 2>&1
timeout 300 python tools/env_profile.py > $E/env_phases_run.txt 2>/dev/null
# the host-fed farm (round 6): group size x lanes sweep, then a kernel trace of the default configuration
HF_CFGS="256:3:60:8 256:5:60:8 512:3:60:8 512:5:100:8 512:5:100 1024:3:40:8" timeout 900 bash tools/hf_sweep.sh > $E/host_fed.txt 2>&1
HF_KIND=c5 HF_CFGS="512:5:60:8" timeout 600 bash tools/hf_sweep.sh >> $E/host_fed.txt 2>&1
( cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/hf_prof -o hf -- \
    python $R/bench.py --host-fed-only c4 --feed-streams 512 --feed-lanes 5 --feed-groups 30 > $E/host_fed_under_rocprof.json 2>/dev/null )
python tools/prof_summary.py kt $R/gpurun_out/hf_prof/hf_results.db > $E/host_fed_kernel_trace_stats.txt 2>&1
# the drop-in library driven by the C application (integration/encode_loop.c), against the unmodified reference beside it
: > $E/encode_loop.txt
for cfg in "1024 0.4 2646000" "4096 0.4 2646000" "65536 0.4 2646000" "1024 0.9 2646000" "65536 0.9 2646000"; do set -- $cfg
  LD_LIBRARY_PATH=build/dropin/ref:build/dropin ./build/dropin/encode_loop $1 $2 $3 write /tmp/el.pkts 2>> $E/encode_loop.txt
  LD_LIBRARY_PATH=build/dropin ./build/dropin/encode_loop $1 $2 $3 check /tmp/el.pkts 2>> $E/encode_loop.txt
done
ls -la $E
