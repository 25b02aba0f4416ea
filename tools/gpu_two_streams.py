"""Scratch: does splitting the batch over independent contexts on their own streams pay?  (Kernels of different stages then
overlap: what one stage leaves idle -- wave slots behind an LDS-full CU, SIMD issue slots behind dependent chains --
another can use.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
nb = 131072
pcm = torch.rand((nb, 2, 2048), device="cuda") - 0.5
want = ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out")
for parts in (1, 2, 3, 4):
    ans = [vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    per = nb // parts
    chunks = [pcm[i * per:(i + 1) * per] for i in range(parts)]
    outs = [a.alloc_outputs(1, per, want) for a in ans]
    for a in ans:
        a.reserve(1, per)
    def step():
        for a, s, c, o in zip(ans, streams, chunks, outs):
            with torch.cuda.stream(s):
                a.analyze(c, outs=o)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%d stream(s): %.3f ms per %d blocks = %.2f M blocks/s" % (parts, dt * 1e3, per * parts, per * parts / dt / 1e6))
    for a in ans:
        a.close()
