"""Turn gpurun_out/profile/* (written on the GPU box by tools/profile.sh) into the committed
summaries under profiles/:  rNN_kernel_trace_stats.txt, rNN_pmc_hbm_traffic.txt, rNN_pmc_traffic.json.

    python tools/make_profiles.py [round-tag, default r01]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profile")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
sys.path.insert(0, ROOT)
import bench as bench_mod  # noqa: E402  (source_hash: the profile is stamped with the sources it was taken from)
NB_PMC = 65536  # tools/prof_run.py 65536 1


def table(path):
    rows = {}
    lines = open(path).read().splitlines()
    for ln in lines[1:]:
        parts = ln.split()
        if len(parts) < 4:
            continue
        n, val, ctr = parts[-1], parts[-2], parts[-3]
        name = ln[:ln.index(ctr)].strip()
        if name.startswith("void k_"):   # instantiated kernels: "void k_transform<11>(...)" -> k_transform
            name = name[5:].split("<")[0]
        rows[name] = float(val)
    return lines, rows


bench = json.loads(open(os.path.join(SRC, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
kt = open(os.path.join(SRC, "kernel_trace_stats.txt")).read().splitlines()
with open(os.path.join(ROOT, "profiles", TAG + "_kernel_trace_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-neighbours\n")
    f.write("# MI355X (gfx950).  %d stereo 2048-blocks per step (C4 full analysis), 6 launches of each stage\n"
            % bench["config"]["blocks_per_gpu"])
    f.write("# kernel = 1 warm-up + 5 timed steps.  Durations in microseconds (rocpd `top_kernels` view).\n")
    f.write("# k_tone_seed / k_tone_chase / k_tone_fold are issued on the library's side stream beside k_noise (fork after the\n")
    f.write("# transform, join before the floor fit); k_noise is persistent and fills the CUs, so they mostly run after it.\n")
    f.write("# source hash %s\n" % bench_mod.source_hash())
    f.write("# bench.py's own line from the same run: %.2f M stereo blocks/s, %.2f ms/step; its HIP-event figure for\n"
            % (bench["value"] / 1e6, bench["ms_per_step"]))
    f.write("# the dominant kernel (k_noise): %.3f ms per launch.\n" % bench["roofline"]["dominant_kernel"]["ms"])
    f.write(kt[0].replace("total_ns", "total_us").replace("avg_ns", "avg_us") + "\n")
    f.write("\n".join(kt[1:]) + "\n")

fl, fetch = table(os.path.join(SRC, "pmc_fetch_size.txt"))
wl, write = table(os.path.join(SRC, "pmc_write_size.txt"))
cal = [k for k in fetch if k.startswith("void at::native::vectorized_elementwise_kernel")][0]
fcal, wcal = fetch[cal] / (1 << 20), write[cal] / (1 << 20)
with open(os.path.join(ROOT, "profiles", TAG + "_pmc_hbm_traffic.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE   (its own pass)   and   rocprofv3 --pmc WRITE_SIZE   (its own pass)\n")
    f.write("#   -- python tools/prof_run.py 65536 1      (one full-analysis step over 65536 stereo blocks, one\n")
    f.write("#      mdct-only call over 131072 frames, one 1 GiB torch copy as the calibration kernel)\n")
    f.write("# Units: KiB per dispatch.  Calibration (vectorized_elementwise_kernel = b.copy_(a), exactly 1 GiB read\n")
    f.write("# and 1 GiB written): FETCH_SIZE reads %.3f of the true bytes (the gfx950 half-count of\n" % fcal)
    f.write("# MI355X_MICROARCH.md \"HBM\"), WRITE_SIZE reads %.3f.  Corrected HBM bytes therefore\n" % wcal)
    f.write("# = 2 x FETCH_SIZE + 1 x WRITE_SIZE.\n\n## FETCH_SIZE\n" + "\n".join(fl) + "\n\n## WRITE_SIZE\n" + "\n".join(wl) + "\n")

per = {}
for k in fetch:
    if k.startswith("k_") and k not in ("k_mdct_only", "k_ampmax"):
        per[k] = {"read_B_per_stereo_block": 2 * fetch[k] * 1024 / NB_PMC,
                  "write_B_per_stereo_block": write[k] * 1024 / NB_PMC}
out = {
    "source_hash": bench_mod.source_hash(),
    "source": "profiles/%s_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
              "65536 stereo blocks; FETCH_SIZE x2 per calibration)" % TAG,
    "workload": "c4",
    "per_kernel": per,
    "total_B_per_stereo_block": sum(v["read_B_per_stereo_block"] + v["write_B_per_stereo_block"] for v in per.values()),
    "mdct_only_B_per_frame": (2 * fetch["k_mdct_only"] + write["k_mdct_only"]) * 1024 / (2 * NB_PMC),
}
json.dump(out, open(os.path.join(ROOT, "profiles", TAG + "_pmc_traffic.json"), "w"), indent=1)
print("total B/stereo block %.0f, mdct-only B/frame %.0f" % (out["total_B_per_stereo_block"], out["mdct_only_B_per_frame"]))
c5 = os.path.join(SRC, "kernel_trace_stats_c5.txt")
if os.path.exists(c5):
    b5 = json.loads(open(os.path.join(SRC, "bench_c5_under_rocprof.json")).read().strip().splitlines()[-1])
    with open(os.path.join(ROOT, "profiles", TAG + "_c5_kernel_trace_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline\n")
        f.write("# %s\n" % b5["config"]["workload"])
        f.write("# bench.py's own line from the same run: %.2f M blocks/s, %.2f ms/step, parity sample %s\n"
                % (b5["value"] / 1e6, b5["ms_per_step"], b5.get("parity_sample")))
        f.write("# source hash %s\n" % bench_mod.source_hash())
        f.write(open(c5).read())
