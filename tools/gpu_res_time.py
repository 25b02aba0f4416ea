"""Scratch: cost of the residue and packet-assembly stages on top of the full analysis."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
name = sys.argv[1] if len(sys.argv) > 1 else "44k_stereo_q4"
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(name), 0)
nb = 65536
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
for want in (("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"),
             ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out", "res_class", "res_entries", "res_count"),
             ("ampmax_out", "packets", "packet_bits")):
    outs = an.alloc_outputs(1, nb, want)
    an.analyze(pcm, outs=outs); torch.cuda.synchronize()
    an.profile(True)
    t0 = time.time()
    for _ in range(5):
        an.analyze(pcm, outs=outs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    ms, runs = an.stage_ms()
    an.profile(False)
    print(name, "packets" if "packets" in want else "residue" if "res_count" in want else "no residue", "ms/step %.3f  Mblocks/s %.3f" % (dt * 1e3, nb / dt / 1e6),
          {k: round(v / runs, 3) for k, v in ms.items()},
          ("mean entries %.0f" % outs["res_count"][:, 1].float().mean().item()) if "res_count" in want else
          ("mean packet %.0f B" % (outs["packet_bits"].float().mean().item() / 8)) if "packets" in want else "")
