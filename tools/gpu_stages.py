"""Scratch: per-stage milliseconds (HIP events) with and without the side stream, and k_noise's phase clock."""
import os, sys, time
os.environ["VAMD_TEST_KNOBS"] = "1"  # (VAMD_NO_OVERLAP is a test knob: vorbis_amd/csrc/vamd_knobs.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pcm = (torch.rand((nb, 2, 2048), device="cuda") - 0.5)
for mode in ("overlap", "serial"):
    if mode == "serial":
        os.environ["VAMD_NO_OVERLAP"] = "1"
    else:
        os.environ.pop("VAMD_NO_OVERLAP", None)
    an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
    outs = an.alloc_outputs(1, nb, ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
    an.analyze(pcm, outs=outs); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        an.analyze(pcm, outs=outs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    an.profile(True)
    for _ in range(3):
        an.analyze(pcm, outs=outs)
    ms, runs = an.stage_ms()
    an.profile(False)
    print(mode, "ms/step %.3f  Mblocks/s %.3f" % (dt * 1e3, nb / dt / 1e6), {k: round(v / runs, 3) for k, v in ms.items() if v})
    if mode == "serial":
        an.debug_cycles(True)
        an.analyze(pcm, outs=outs); torch.cuda.synchronize()
        c = an.debug_cycles(False, read=True)
        nz = c[1]
        tot = float(sum(nz)) or 1.0
        print("k_noise phases (share of stopwatch ticks):", [round(float(x) / tot, 3) for x in nz[:9]])
    an.close()
