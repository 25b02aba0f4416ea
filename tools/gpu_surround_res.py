"""Scratch: residue stage of the 5.1 layout: entries per block and phase clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
name = sys.argv[1] if len(sys.argv) > 1 else "44k_51_q3"
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(name), 0)
nb = 8192
ch = an.channels
pcm = (torch.rand((nb, ch, 2048), device="cuda") - 0.5)
outs = an.alloc_outputs(1, nb, ("ampmax_out", "res_class", "res_entries", "res_count"))
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
cnt = outs["res_count"].reshape(nb, -1, 2).float().mean(0)
print(name, "mean (classes, entries) per submap:", cnt.tolist(), "capacity", an.residue_capacity(1))
an.debug_cycles(True)
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
c = an.debug_cycles(False, read=True)
print("residue phase ticks (classify+offsets, search):", c[4][8:12].tolist())
