"""Scratch: parity spot-check + per-stage timing on the GPU (used while optimising)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vorbis_amd
from tests import checker
ALL = ("mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct", "posts", "post_valid", "ilogmask",
       "iwork", "nonzero", "local_ampmax", "ampmax_out")
for name in ("44k_stereo_q4", "44k_stereo_q1"):
    an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(name), 0)
    chk = checker.Checker(name)
    rng = np.random.default_rng(5)
    pcm = ((rng.random((24, 2, 2048), dtype=np.float32) - 0.5) * 2 * np.array([0.5, 0.01, 1.0])[np.arange(24) % 3, None, None]).astype(np.float32)
    pcm[7, 1] = 0; pcm[9] = 0
    outs = an.analyze(torch.from_numpy(pcm).cuda(), want=ALL); torch.cuda.synchronize()
    bad = sum(checker.compare_block(chk.tap_block(pcm[b]), {k: v[b].cpu().numpy() for k, v in outs.items()}, 29, verbose=True) for b in range(24))
    print(name, "bad tensors:", bad)
    ps = (rng.random((16, 2, 256), dtype=np.float32) - 0.5).astype(np.float32)
    outs = an.analyze(torch.from_numpy(ps).cuda(), W=0, lW=0, nW=0, blocktype=0, want=ALL); torch.cuda.synchronize()
    bad = sum(checker.compare_block(chk.tap_block(ps[b], 0, 0, 0, 0), {k: v[b].cpu().numpy() for k, v in outs.items()}, an.posts[0], verbose=True) for b in range(16))
    print(name, "short bad tensors:", bad)
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
an.profile(True)
t0 = time.time()
for _ in range(5):
    an.analyze(pcm, outs=outs)
torch.cuda.synchronize()
dt = (time.time() - t0) / 5
ms, runs = an.stage_ms()
print("nb", nb, "ms/step %.3f" % (dt * 1e3), "Mblocks/s %.3f" % (nb / dt / 1e6), {k: round(v / runs, 3) for k, v in ms.items()})
an.profile(False)
x = torch.rand((nb * 2, 2048), device="cuda") - 0.5
y = an.mdct_forward(1, x); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    an.mdct_forward(1, x, out=y)
torch.cuda.synchronize()
dt = (time.time() - t0) / 5
print("mdct-only Mframes/s %.2f  GB/s %.1f" % (nb * 2 / dt / 1e6, nb * 2 * 12288 / dt / 1e9))
# phase stopwatch (ticks per wave, averaged)
an.debug_cycles(True)
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
cyc = an.debug_cycles(False, read=True)
names = ["transform", "noise", "tone", "floor", "couple"]
for k in range(5):
    nw = nb * (1 if k == 4 else 2)
    print(names[k], "kcycles/wave per phase:", [round(float(c) / nw / 1e3, 1) for c in cyc[k][:8]], "sum", round(float(cyc[k].sum()) / nw / 1e3, 1))
# per-block host API latency and a PCIe-inclusive batch rate
blk = (np.random.default_rng(1).random((2, 2048), dtype=np.float32) - 0.5)
an.analyze_block(blk)
t0 = time.time()
for _ in range(300):
    an.analyze_block(blk)
print("vamd_analyze_block: %.1f us/block" % ((time.time() - t0) / 300 * 1e6))
nbp = 16384
hp = torch.empty((nbp, 2, 2048), dtype=torch.float32).pin_memory(); hp.uniform_(-0.5, 0.5)
want = ("mdct", "posts", "post_valid", "iwork", "nonzero", "ampmax_out")
outs2 = an.alloc_outputs(1, nbp, want)
host_out = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in outs2.items()}
dp = torch.empty((nbp, 2, 2048), device="cuda")
def roundtrip():
    dp.copy_(hp, non_blocking=True)
    an.analyze(dp, outs=outs2)
    for k in outs2:
        host_out[k].copy_(outs2[k], non_blocking=True)
    torch.cuda.synchronize()
roundtrip()
t0 = time.time()
for _ in range(3):
    roundtrip()
dt = (time.time() - t0) / 3
print("PCIe-inclusive (pinned H2D pcm + analysis + D2H mdct/posts/iwork): %.3f M stereo blocks/s" % (nbp / dt / 1e6))
