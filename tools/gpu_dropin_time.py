"""Scratch: wall time of a whole-stream encode through the hybrid libvorbis (per-block GPU calls) vs the CPU reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
for ch, q in ((2, 0.4), (6, 0.3)):
    rng = np.random.default_rng(1)
    x = ((rng.random((ch, int(44100 * secs)), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    for hybrid in (False, True):
        e = ref.RefEncoder(ch, 44100, q, hybrid=hybrid)
        e.encode_stream(x[:, :44100])          # warm-up (context creation on the GPU side)
        e = ref.RefEncoder(ch, 44100, q, hybrid=hybrid)
        t0 = time.time()
        blocks = e.encode_stream(x)
        dt = time.time() - t0
        print("%d ch q%.1f %-6s %.2f s for %.0f s of audio (%.1fx real time), %d blocks, %.0f us per block"
              % (ch, q, "hybrid" if hybrid else "cpu", dt, secs, secs / dt, len(blocks), dt / len(blocks) * 1e6))
