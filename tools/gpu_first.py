"""First contact with the GPU: smoke parity + per-stage timings (scratch tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
import vorbis_amd
from tests import checker

print("device:", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
g.smoke()

# mixed stream blocks (short/long/transition) against the live reference
from oracle import ref
for name in ("44k_stereo_q9", "44k_stereo_q1", "44k_mono_q5"):
    ch, rate, q = checker.SETUPS[name]
    rng = np.random.default_rng(7)
    frames = 44100 * 2
    t = np.arange(frames)
    gate = np.where((t % 11025) < 1102, 0.5, 0.0005).astype(np.float32)
    pcm = ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * gate).astype(np.float32)
    blocks = ref.RefEncoder(ch, rate, q).encode_stream(pcm)
    an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(name), 0)
    chk = checker.Checker(name)
    want = ("mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct", "posts", "post_valid", "ilogmask",
            "iwork", "nonzero", "local_ampmax", "ampmax_out")
    nbad = 0
    for W in (0, 1):
        sel = [b for b in blocks if b["W"] == W]
        if not sel:
            continue
        P = torch.from_numpy(np.stack([b["pcm"] for b in sel])).cuda()
        dv = lambda k, dt: torch.tensor([b[k] for b in sel], dtype=dt).cuda()
        outs = an.analyze(P, W=W, lW=dv("lW", torch.int32), nW=dv("nW", torch.int32),
                          blocktype=dv("blocktype", torch.int32), ampmax_in=dv("ampmax_in", torch.float32), want=want)
        torch.cuda.synchronize()
        for i, b in enumerate(sel):
            r = chk.tap_block(b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])
            nbad += checker.compare_block(r, {k: v[i].cpu().numpy() for k, v in outs.items()}, an.posts[W], verbose=(nbad < 5))
    print(name, "blocks", len(blocks), "short", sum(1 for b in blocks if b["W"] == 0), "bad tensors", nbad)

# timings
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
for nb in (4096, 32768):
    pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
    outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
    an.analyze(pcm, outs=outs); torch.cuda.synchronize()
    an.profile(True)
    t0 = time.time()
    for _ in range(3):
        an.analyze(pcm, outs=outs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    ms, runs = an.stage_ms()
    print("nb", nb, "full ms/step %.3f" % (dt * 1e3), "blocks/s %.3e" % (nb / dt), {k: round(v / runs, 3) for k, v in ms.items()})
    an.profile(False)
    x = torch.rand((nb * 2, 2048), device="cuda") - 0.5
    y = an.mdct_forward(1, x); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        an.mdct_forward(1, x, out=y)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print("  mdct-only frames/s %.3e  GB/s %.1f" % (nb * 2 / dt, nb * 2 * 12288 / dt / 1e9))
