"""Scratch: stopwatch ticks of the tone kernels per wave (serial mode)."""
import os, sys
os.environ["VAMD_TEST_KNOBS"] = "1"  # (VAMD_NO_OVERLAP is a test knob: vorbis_amd/csrc/vamd_knobs.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VAMD_NO_OVERLAP"] = "1"
import torch
import vorbis_amd
nb = 32768
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
an.debug_cycles(True)
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
c = an.debug_cycles(False, read=True)
t = c[2]
print("tone slots (ticks):", [int(x) for x in t[:6]])
print("seed: init %.0f scatter %.0f per wave; chase %.0f per wave (64 blocks); fold: paint %.0f fold %.0f per wave" % (
    t[0] / (2 * nb), t[1] / (2 * nb), t[2] / (2 * nb / 64), t[3] / (2 * nb), t[4] / (2 * nb)))
