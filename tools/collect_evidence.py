"""gpurun_out/evidence/* (tools/evidence.sh) -> profiles/rNN_*: the text evidence of a round, each file stamped with the
source hash it was taken from.  Run after tools/make_profiles.py rNN.   python tools/collect_evidence.py r06"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "evidence")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
sys.path.insert(0, ROOT)
import bench  # noqa: E402
ran = open(os.path.join(E, "source_hash.txt")).read().strip()
now = bench.source_hash()
if ran != now:
    print("WARNING: evidence was taken from sources %s, the tree is at %s" % (ran, now))


def stamp(src, dst, head):
    if not os.path.exists(src):
        print("missing", src)
        return
    with open(os.path.join(ROOT, "profiles", TAG + "_" + dst), "w") as f:
        f.write("# %s\n# source hash %s (tools/evidence.sh on one MI355X)\n" % (head, ran))
        f.write(open(src).read())
    print("profiles/%s_%s" % (TAG, dst))


stamp(os.path.join(E, "batcher.txt"), "batcher.txt", "many encoder threads through the hybrid libvorbis (VAMD_BATCH) against the unmodified reference: tools/batcher_sweep.sh")
stamp(os.path.join(E, "block_path.txt"), "block_path.txt", "the per-block path (vamd_encode_block: one stereo block, host PCM in, packet out): tools/kt_block.sh, gpu_block_latency.py, gpu_block_phases.py")
stamp(os.path.join(E, "lookahead.txt"), "lookahead.txt", "one thread, one stream, samples per vorbis_analysis_wrote() varied: tools/gpu_lookahead_bench.py")
stamp(os.path.join(E, "soak_lookahead.txt"), "soak_lookahead.txt", "tools/soak_lookahead.py 60")
stamp(os.path.join(E, "alt_paths.txt"), "alt_paths.txt", "the GPU suite with every test knob turned the other way: tools/alt_paths.sh")
stamp(os.path.join(E, "host_fed.txt"), "host_fed.txt", "the host-fed farm (vamd_feed): streams per group x lanes x groups [x GPU_MAX_HW_QUEUES]; 131072-frame stereo streams, s16 from pinned host memory in, packets in host memory out: tools/hf_sweep.sh")
stamp(os.path.join(E, "host_fed_kernel_trace_stats.txt"), "host_fed_kernel_trace_stats.txt", "rocprofv3 --kernel-trace --stats -- python bench.py --host-fed-only c4 --feed-streams 512 --feed-lanes 5 --feed-groups 30 (GPU_MAX_HW_QUEUES=8); durations of kernels of five groups in flight overlap")
stamp(os.path.join(E, "encode_loop.txt"), "encode_loop.txt", "integration/encode_loop.c (the call sequence of examples/encoder_example.c:140-236, 60 s of stereo 16-bit noise) on build/dropin/ref (write) and on the drop-in (check): READ, quality, blocks/s")
stamp(os.path.join(E, "res_pack.txt"), "res_pack.txt", "the residue search and the packet assembly of 65 536 stereo long blocks (k_residue_chunks, k_pack_waves): tools/res_profile.py (stage ms per batch by HIP events, phase stopwatch), the road not taken beside it, then tools/pmc_res.sh (counters per wave: k_residue_chunks' waves take 8 blocks each, k_pack_waves' 10.7; FETCH_SIZE / WRITE_SIZE in counted KiB per wave, FETCH x 2 = bytes on gfx950)")
stamp(os.path.join(E, "env_phases_run.txt"), "env_phases_run.txt", "tools/env_profile.py")
stamp(os.path.join(E, "pytest_gpu.txt"), "pytest_gpu.txt", "python -m pytest tests -m gpu -q")
soak = os.path.join(ROOT, "gpurun_out", "soak.txt")
if os.path.exists(soak):
    shutil.copy(soak, os.path.join(ROOT, "profiles", TAG + "_soak.txt"))
    print("profiles/%s_soak.txt" % TAG)
# the bench lines, one row per workload + the full JSON of the default line
rows = []
for w in ("c4", "c5", "c3", "c2"):
    p = os.path.join(E, "bench_%s.json" % w)
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append("%s: no line (%r)" % (w, e))
        continue
    r = d["roofline"]
    v = r.get("valu") or {}
    rows.append("%s: %.2f M %s, %.3f ms/step over %.2f s, frac %.4f (alg %.3f GB/step), traffic %s, valu frac %s lanes %s, parity %s"
                % (w, d["value"] / 1e6, d["unit"], d["ms_per_step"], d.get("timed_region_s", 0.0), r["frac"], r.get("alg_bytes_per_step", 0) / 1e9,
                   ("%.2f GB" % (r["traffic"] / 1e9)) if r.get("traffic") else r.get("traffic_source"),
                   ("%.3f" % v["frac_valu"]) if v.get("frac_valu") else v.get("note", v.get("issue_ms_per_step")),
                   ("%.3f" % v["lane_utilisation"]) if v.get("lane_utilisation") else None, d.get("parity_sample")))
    if w == "c4":
        for k, x in (d.get("host_fed") or {}).items():
            rows.append("   host_fed %s: %s" % (k, x.get("error") or ("%.2f M blocks/s over %.2f s, up %.1f GB/s sustained (%.1f while copying), %s, parity %s"
                        % (x["value"] / 1e6, x["seconds"], x["pcie_GBps"]["up_sustained"], x["pcie_GBps"]["up_while_copying"], x["output"], x.get("parity_sample")))))
        for k, x in (d.get("workloads") or {}).items():
            rows.append("   in the default line, %s: %s" % (k, x.get("error") or ("%.2f M, %.3f ms/step, frac %.4f" % (x["value"] / 1e6, x["ms_per_step"], x["roofline"]["frac"]))))
        cb = d.get("cpu_baseline") or {}
        rows.append("   cpu_baseline: %s %s on %s workers (%s)" % (cb.get("value"), cb.get("unit"), cb.get("cores"), cb.get("kind")))
        json.dump(d, open(os.path.join(ROOT, "profiles", TAG + "_bench_default_line.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", TAG + "_bench_lines.txt"), "w") as f:
    f.write("# bench.py lines of tools/evidence.sh (python bench.py --workload W), source hash %s\n" % ran)
    f.write("\n".join(rows) + "\n")
print("\n".join(rows))
