"""Scratch: throughput of the block-switching detector (vamd_envelope_search_batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_amd
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 256
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
steps = int(secs * 44100) // 64 - 1
ln = (steps - 1) * 64 + 128
pcm = torch.rand((ns, 2, ln), device="cuda") - 0.5
ret, st = an.envelope_search_batch(pcm, steps)
torch.cuda.synchronize()
t0 = time.time()
R = 3
for _ in range(R):
    st.zero_()
    an.envelope_search_batch(pcm, steps, states=st, ret=ret)
torch.cuda.synchronize()
dt = (time.time() - t0) / R
print("streams %d x %.0f s: %d steps each, %.3f ms, %.1f M stereo steps/s = %.0f x real time aggregate; single-stream serial walk bound below"
      % (ns, secs, steps, dt * 1e3, ns * steps / dt / 1e6, ns * secs / dt))
one = pcm[:1].contiguous()
ret1, st1 = an.envelope_search_batch(one, steps)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(R):
    st1.zero_()
    an.envelope_search_batch(one, steps, states=st1, ret=ret1)
torch.cuda.synchronize()
dt = (time.time() - t0) / R
print("one stream of %.0f s: %.3f ms = %.0f x real time" % (secs, dt * 1e3, secs / dt))
