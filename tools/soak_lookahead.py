"""Soak of the look-ahead inside one stream (integration/mapping0_vamd.c): random streams, every write a pseudo-random
1 .. W samples, every pull a pseudo-random 0 .. D blocks (oracle/ref_harness.c's jitter), the reference and the hybrid
driven alike; every packet and every ampmax compared.  Usage: python tools/soak_lookahead.py [runs] [out.txt]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
L = ref.lib(hybrid=True)


def stats():
    h, m, b = C.c_long(), C.c_long(), C.c_long()
    L.vamd_ahead_stats(C.byref(h), C.byref(m), C.byref(b))
    return np.array([h.value, m.value, b.value])


def stream(rng, ch, seconds, kind):
    frames = int(44100 * seconds)
    x = rng.random((ch, frames), dtype=np.float32) - 0.5
    t = np.arange(frames)
    if kind == "gated":
        x *= 2 * np.where((t % int(rng.integers(4000, 14000))) < int(rng.integers(300, 1500)), 0.5, 0.0005).astype(np.float32)
    elif kind == "s16":
        x = np.round(x * 32767).astype(np.int16).astype(np.float32) / 32768.0
    elif kind == "tone":
        x = (0.3 * np.sin(2 * np.pi * rng.uniform(100, 8000) * t / 44100.0)[None, :] + 0.01 * x).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


print("# tools/soak_lookahead.py %d: random write sizes and pulls, reference against hybrid, packet by packet" % runs, file=out)
rng = np.random.default_rng(2026)
t0 = time.time()
blocks = bad = 0
total = np.zeros(3, dtype=np.int64)
for r in range(runs):
    ch = int(rng.choice([1, 2, 2, 2, 6]))
    managed = (ch == 2 and rng.random() < 0.25)
    q = float(rng.choice([0.1, 0.3, 0.4, 0.6, 0.9])) if ch != 6 else 0.3
    kind = str(rng.choice(["gated", "s16", "tone", "gated"]))
    write = int(rng.choice([1500, 5000, 12000, 40000, 100000]))
    drain = int(rng.choice([0, 1, 3, 10, 50]))
    seed = int(rng.integers(1, 1 << 30))
    pcm = stream(rng, ch, float(rng.uniform(3, 9)), kind)
    args = (ch, 44100) if managed else (ch, 44100, q)
    kw = dict(managed=(-1, int(rng.choice([96000, 128000, 192000])), -1)) if managed else {}
    s0 = stats()
    want = ref.RefEncoder(*args, **kw).encode_stream(pcm, write_frames=write, drain=drain, jitter=seed)
    got = ref.RefEncoder(*args, hybrid=True, **kw).encode_stream(pcm, write_frames=write, drain=drain, jitter=seed)
    d = stats() - s0
    total += d
    miss = 0 if len(want) == len(got) else 1
    for a, b in zip(want, got):
        if (a["lW"], a["W"], a["nW"], a["blocktype"]) != (b["lW"], b["W"], b["nW"], b["blocktype"]) or a["packet"] != b["packet"] \
                or np.float32(a["ampmax_out"]).tobytes() != np.float32(b["ampmax_out"]).tobytes():
            miss += 1
    blocks += len(want)
    bad += miss
    print("run %3d: %d ch %s %-5s writes <= %6d pulls <= %2d: %5d blocks, look-ahead hits / stale plans / batches %5d / %3d / %4d, %d mismatches"
          % (r, ch, ("managed" if managed else "q %.1f" % q), kind, write, drain, len(want), d[0], d[1], d[2], miss), file=out, flush=True)
print("LOOKAHEAD SOAK %s %d blocks, %d mismatches; hits / stale plans / batches %d / %d / %d; %.0f s"
      % ("OK" if bad == 0 else "FAILED", blocks, bad, total[0], total[1], total[2], time.time() - t0), file=out, flush=True)
sys.exit(1 if bad else 0)
