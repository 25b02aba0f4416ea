"""Code-object resource summary of the product kernels: VGPRs / SGPRs / spills / scratch / occupancy as hipcc reports
them (-Rpass-analysis=kernel-resource-usage), one line per kernel whose demangled name matches the pattern.

    python tools/kernel_resources.py ['k_noise|k_floor']
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
extra = sys.argv[2:]   # further compiler flags, e.g. -DVAMD_NOISE_NO_PREFETCH
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vorbis_amd", "csrc"),
       "-Rpass-analysis=kernel-resource-usage"] + extra + ["-c", os.path.join(ROOT, "vorbis_amd", "csrc", "vamd_hip.hip"), "-o", "/tmp/vamd_res.o"]
err = subprocess.run(cmd, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=900).stderr
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
if not rows:  # the compile failed (c++filt without arguments would wait on stdin)
    sys.exit("no kernels reported:\n" + "\n".join(l for l in err.splitlines() if "error" in l)[:4000])
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
print("%-44s %5s %5s %7s %7s %7s %4s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "occ"))
for r, nm in zip(rows, names):
    nm = re.sub(r"\(.*$", "", nm).replace("void ", "")
    if pat.search(nm):
        print("%-44s %5s %5s %7s %7s %7s %4s" % (nm, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                                 r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
