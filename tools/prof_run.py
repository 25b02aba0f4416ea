"""Workload for rocprofv3: a few full-analysis steps + mdct-only + a calibration copy of known size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
pcm = torch.rand((nb, 2, 2048), device="cuda") - 0.5
outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
an.reserve(1, nb)
for _ in range(steps):
    an.analyze(pcm, outs=outs)
x = pcm.reshape(nb * 2, 2048)
y = torch.empty((nb * 2, 1024), device="cuda")
for _ in range(steps):
    an.mdct_forward(1, x, out=y)
# calibration: exactly 1 GiB read + 1 GiB written by the library's own named kernel (k_calib_copy, 16 bytes per lane);
# tools/make_profiles.py derives the FETCH_SIZE / WRITE_SIZE factors from ITS rows
a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
for _ in range(steps):
    an.calib_copy(b, a)
torch.cuda.synchronize()
