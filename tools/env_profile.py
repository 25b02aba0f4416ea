"""The block-switching detector's spectrum kernel phase by phase: the in-kernel stopwatch of
k_env_spectrum over a batch of streams -- slots: 0 window, 1 fold, 2-4 butterfly / bit-reverse trips of the 128-point
MDCT, 5 its last trip, 6 near-DC term + dB pairs, 7 the wait between items.

    python tools/env_profile.py [streams] [samples]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q9"), 0)
torch.manual_seed(0)
streams = torch.rand((ns, an.channels, samples), device="cuda") - 0.5
win, step = an.envelope_geometry()
steps = (samples - win) // step + 1
ret, st = an.envelope_search_batch(streams, steps)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    st.zero_()
    an.envelope_search_batch(streams, steps, states=st, ret=ret)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("%d streams x %d steps: %.3f ms per call, %.1f M stereo steps/s" % (ns, steps, dt * 1e3, ns * steps / dt / 1e6))
an.debug_cycles(True)
st.zero_()
an.envelope_search_batch(streams, steps, states=st, ret=ret)
torch.cuda.synchronize()
c = an.debug_cycles(False, read=True)
items = ns * an.channels * ((steps + 3) // 4)
print("k_env_spectrum cycles per item (four steps of one channel, one wave) by phase:", [round(float(x) / items) for x in c[0][:8]])
