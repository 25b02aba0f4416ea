"""Scratch: k_noise's phase stopwatch (kcycles per wave and phase) and the stage times, for the environment as set."""
import os, sys, time
os.environ["VAMD_TEST_KNOBS"] = "1"  # (VAMD_NO_OVERLAP is a test knob: vorbis_amd/csrc/vamd_knobs.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
os.environ["VAMD_NO_OVERLAP"] = "1"
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
an.profile(True)
for _ in range(3):
    an.analyze(pcm, outs=outs)
ms, runs = an.stage_ms()
an.profile(False)
print({k: round(v / runs, 3) for k, v in ms.items() if v})
an.debug_cycles(True)
an.analyze(pcm, outs=outs); torch.cuda.synchronize()
c = an.debug_cycles(False, read=True)
names = ["transform", "noise", "tone", "floor", "couple"]
for k in range(5):
    nw = nb * 2 * (4 if k == 1 else 1) / (2 if k == 4 else 1)
    print(names[k], "kcycles/wave per phase:", [round(float(x) / nw / 1e3, 2) for x in c[k][:8]], "sum", round(float(c[k].sum()) / nw / 1e3, 1))
