"""Workload for rocprofv3 counter passes: BASELINE config 5 exactly as `bench.py --workload c5` runs it (the bench's own
StreamRunner: plan on the device, gather, both size classes, per-stream ampmax chains), `steps` steps, then a
calibration copy of known size.  Prints the step's block counts and algorithmic bytes as one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
a = bench.parse(["--workload", "c5"])
a.setup = a.setup or "44k_stereo_q9"
import vorbis_amd
blob = vorbis_amd.default_setup_blob(a.setup)
R = bench.StreamRunner(a, blob, torch.device("cuda:0"), 0, 1)   # (its constructor runs one step: allocation + warm-up)
for _ in range(steps):
    R.step()
R.sync()
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
for _ in range(steps):
    R.an.calib_copy(y, x)     # the library's named copy kernel (k_calib_copy): 1 GiB in, 1 GiB out
torch.cuda.synchronize()
print(json.dumps({"short_blocks": int(R.plan.nblocks[0]), "long_blocks": int(R.plan.nblocks[1]), "alg_bytes": R.alg_bytes(),
                  "dispatches_per_kernel": steps + 1}))
