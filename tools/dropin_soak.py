"""Extended drop-in check: whole streams of several kinds through the hybrid libvorbis (GPU analysis + detector) against
the pure CPU reference, every packet.  python tools/dropin_soak.py [seconds per stream]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
rng = np.random.default_rng(4242)
def stream(ch, kind, rate):
    n = int(rate * secs)
    t = np.arange(n)
    if kind == "gated":
        x = (rng.random((ch, n)) - 0.5) * 2 * np.where((t % 9000) < 900, 0.5, 0.0005)
    elif kind == "music":
        f0 = rng.uniform(0.002, 0.01)
        env = np.exp(-(t % 22050) / 6000.0)
        s = sum(np.sin(2 * np.pi * f0 * h * t) / h for h in range(1, 9)) * env
        x = np.round(0.4 * s[None, :] * rng.uniform(0.6, 1.0, (ch, 1)) * 32767) / 32768 + (rng.random((ch, n)) - 0.5) * 1e-3
    elif kind == "loud":
        x = (rng.random((ch, n)) - 0.5) * 3.0 + 2.0 * np.sin(t * 0.03)[None, :]
    elif kind == "quiet":
        x = (rng.random((ch, n)) - 0.5) * 1e-4
        x[:, n // 3: n // 3 + 2000] = 0.0
    else:
        x = (rng.random((ch, n)) - 0.5) * 0.3
    return np.ascontiguousarray(x, np.float32)
bad = total = 0
t0 = time.time()
for ch, rate, q in ((2, 44100, 0.4), (2, 44100, 0.9), (2, 44100, 0.1), (1, 44100, 0.5), (6, 44100, 0.3), (2, 22050, 0.3), (2, 96000, 0.6)):
    for kind in ("gated", "music", "loud", "quiet", "noise"):
        x = stream(ch, kind, rate)
        want = ref.RefEncoder(ch, rate, q).encode_stream(x)
        got = ref.RefEncoder(ch, rate, q, hybrid=True).encode_stream(x)
        b = len(want) != len(got)
        for a, g in zip(want, got):
            b += a["packet"] != g["packet"] or (a["lW"], a["W"], a["nW"], a["blocktype"]) != (g["lW"], g["W"], g["nW"], g["blocktype"])
        total += len(want)
        bad += b
        print("%d ch %d Hz q %.1f %-6s %5d blocks (%d short)  mismatches %d" % (ch, rate, q, kind, len(want), sum(1 for w in want if w["W"] == 0), b), flush=True)
print("DROPIN SOAK", "FAILED" if bad else "OK", total, "blocks", bad, "mismatches, %.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
