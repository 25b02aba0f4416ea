"""Scratch: the in-kernel phase stopwatch of every stage (shares of each kernel's ticks), serial mode."""
import os, sys
os.environ["VAMD_TEST_KNOBS"] = "1"  # (VAMD_NO_OVERLAP is a test knob: vorbis_amd/csrc/vamd_knobs.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VAMD_NO_OVERLAP"] = "1"
import torch
import vorbis_amd
nb = 16384
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(sys.argv[1] if len(sys.argv) > 1 else "44k_stereo_q4"), 0)
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
an.debug_cycles(True)
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
c = an.debug_cycles(False, read=True)
for name, row in zip(("transform", "noise", "tone", "floor", "couple"), c):
    tot = float(row.sum()) or 1.0
    print("%-10s total ticks %12d  shares %s" % (name, int(row.sum()), [round(float(x) / tot, 3) for x in row[:10]]))
