"""Scratch: does splitting a batch over two contexts/streams overlap usefully?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
K = int(os.environ.get("K", "2"))
nb = 65536
ans = [vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
parts = pcm.chunk(K)
outs = [a.alloc_outputs(1, p.shape[0], ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out")) for a, p in zip(ans, parts)]
def step():
    for a, p, o, s in zip(ans, parts, outs, streams):
        with torch.cuda.stream(s):
            a.analyze(p, outs=o)
step(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / 5
print(os.environ.get("TAG", ""), "K", K, "ms/step %.3f" % (dt * 1e3), "Mblocks/s %.3f" % (nb / dt / 1e6))
