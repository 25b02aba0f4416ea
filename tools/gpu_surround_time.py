"""Scratch: throughput of the 5.1 layout (six channels per block), analysis only and PCM -> packets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
name = sys.argv[1] if len(sys.argv) > 1 else "44k_51_q3"
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(name), 0)
nb = 16384
pcm = (torch.rand((nb, 6, 2048), device="cuda") - 0.5)
pcm[:, 1] = 0.8 * pcm[:, 0] + 0.2 * pcm[:, 1]
for want in (("mdct", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"), ("ampmax_out", "packets", "packet_bits")):
    outs = an.alloc_outputs(1, nb, want)
    an.analyze(pcm, outs=outs); torch.cuda.synchronize()
    an.profile(True)
    t0 = time.time()
    for _ in range(3):
        an.analyze(pcm, outs=outs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    ms, runs = an.stage_ms()
    an.profile(False)
    print(name, "packets" if "packets" in want else "analysis", "ms/step %.3f  M blocks/s %.3f (= %.2f M channel-blocks/s)"
          % (dt * 1e3, nb / dt / 1e6, 6 * nb / dt / 1e6), {k: round(v / runs, 3) for k, v in ms.items() if v},
          ("mean packet %.0f B" % (outs["packet_bits"].float().mean().item() / 8)) if "packets" in want else "")
