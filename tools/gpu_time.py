"""Scratch: whole-pipeline throughput only (no parity check); used for launch-shape experiments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    an.analyze(pcm, outs=outs)
torch.cuda.synchronize()
dt = (time.time() - t0) / 5
print(os.environ.get("TAG", ""), "nb", nb, "ms/step %.3f" % (dt * 1e3), "Mblocks/s %.3f" % (nb / dt / 1e6))
