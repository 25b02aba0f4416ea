"""Turn the outputs of one tools/evidence.sh run (gpurun_out/evidence/, gpurun_out/soak.txt, gpurun_out/profile/) into the
tracked profiles/rNN_* files: tools/make_profiles.py for the counter / trace passes, and the text files below with the
run's source hash in their headers (the explanatory header lines of an existing file are kept).
Usage: python tools/refresh_profiles.py r05"""
import json, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
E = os.path.join(R, "gpurun_out", "evidence")
P = os.path.join(R, "profiles")
H = open(os.path.join(E, "source_hash.txt")).read().strip()
HASH = re.compile(r"\b[0-9a-f]{16}\b")


def header(path):
    out = []
    if os.path.exists(path):
        for l in open(path):
            if not l.startswith("#"):
                break
            out.append(HASH.sub(H, l))
    return out


subprocess.check_call([sys.executable, os.path.join(R, "tools", "make_profiles.py"), tag], cwd=R)
open(os.path.join(P, tag + "_soak.txt"), "w").write(open(os.path.join(R, "gpurun_out", "soak.txt")).read())
for name in ("alt_paths", "batcher", "block_path", "lookahead", "soak_lookahead"):
    dst = os.path.join(P, "%s_%s.txt" % (tag, name))
    h = header(dst)
    body = open(os.path.join(E, name + ".txt")).read().split("\n")
    if name == "alt_paths":
        h = h[:1] + ["# source_hash %s\n" % H]
        body = [l for l in body if not l.startswith("#")]
    if name == "batcher":
        h = [l for l in h if not l.startswith("# host:")]
    if name == "soak_lookahead":
        h = ["# source hash %s (csrc + include; the binding, integration/mapping0_vamd.c, at this commit)\n" % H]
    if name == "block_path":
        body = [re.sub(r"\(ticks / 2.4 GHz\)", "", l) for l in body]
    open(dst, "w").write("".join(h) + "\n".join(body))

lines = ["# bench.py lines of the evidence run (tools/evidence.sh), round 5, one MI355X, source hash %s\n" % H,
         "# workload   value                     ms/step   roofline.frac   shader clock (measured)   parity_sample mismatches   stage ms\n"]
for w in ("c4", "c5", "c3", "c2"):
    d = json.loads(open(os.path.join(E, "bench_%s.json" % w)).read().strip().splitlines()[-1])
    sc = d.get("shader_clock") or {}
    r = d["roofline"]
    st = {k: round(v, 3) for k, v in r.get("kernels_ms_per_step", {}).items()}
    unit = {"c2": "frames/s"}.get(w, "stereo blocks/s")
    lines.append("%s  %.2f M %s  %.3f  %.4f  %s  %s  %s\n" % (w, d["value"] / 1e6, unit, d["ms_per_step"], r["frac"],
                 ("%.3f GHz" % sc["ghz"]) if sc.get("ghz") else "-", d.get("parity_sample", {}).get("mismatches"), st))
    if w == "c4":
        lines.append("   c4's roofline: traffic %s (%s); valu %s\n" % (r.get("traffic"), r.get("traffic_source"), json.dumps(r.get("valu"))[:300]))
        for k, v in (d.get("workloads") or {}).items():
            lines.append("   (c4's line also carries) %s  %.2f M  %.3f ms  frac %.4f  parity mismatches %s\n"
                         % (k, v["value"] / 1e6, v["ms_per_step"], v["roofline"]["frac"], v.get("parity_sample", {}).get("mismatches")))
        lines.append("   neighbours: %s\n" % json.dumps({k: round(v["value"]) for k, v in (d.get("neighbours") or {}).items()}))
        lines.append("   cpu_baseline: %s\n" % json.dumps(d.get("cpu_baseline"))[:600])
open(os.path.join(P, tag + "_bench_lines.txt"), "w").write("".join(lines))
print("".join(lines))
print(open(os.path.join(E, "pytest_gpu.txt")).read(), open(os.path.join(E, "soak_tail.txt")).read())
