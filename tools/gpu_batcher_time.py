"""Scratch: aggregate throughput of N encoder threads through the hybrid libvorbis, per-state contexts against the
batcher (VAMD_BATCH).  Run each mode in its own process: python tools/gpu_batcher_time.py N [seconds]."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rng = np.random.default_rng(1)
x = ((rng.random((2, int(44100 * secs)), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
hyb = os.environ.get("VAMD_CPU_ONLY") is None
keeper = ref.RefEncoder(2, 44100, 0.4, hybrid=hyb)   # opens the GPU side (context / batcher) before the clock starts
keeper.encode_stream(x[:, :22050])                   # ... and keeps it open while the timed encoders run
encs = [ref.RefEncoder(2, 44100, 0.4, hybrid=hyb) for _ in range(N)]
nb = [0] * N
def work(k):
    nb[k] = len(encs[k].encode_stream(x))
th = [threading.Thread(target=work, args=(k,)) for k in range(N)]
c0 = os.times()
t0 = time.time()
[t.start() for t in th]
[t.join() for t in th]
wall = time.time() - t0
c1 = os.times()
print("host CPU: %.3f ms user + %.3f ms system per block" % (1e3 * (c1.user - c0.user) / max(sum(nb), 1), 1e3 * (c1.system - c0.system) / max(sum(nb), 1)))
if hyb and os.environ.get("VAMD_BATCH"):
    import ctypes as C
    a, b, t = C.c_long(0), C.c_long(0), C.c_double(0)
    ref.lib(hybrid=True).vamd_batch_stats(C.byref(a), C.byref(b), C.byref(t))
    print("batches %d blocks %d -> %.1f blocks per batch, %.2f ms wall per batch, %.2f ms inside the GPU call" % (a.value, b.value, b.value / max(a.value, 1), 1e3 * wall / max(a.value, 1), 1e3 * t.value / max(a.value, 1)))
print("mode %s  threads %d  blocks %d  wall %.2f s  -> %.0f blocks/s, %.1fx real time per stream, %.0fx aggregate" % (
    "cpu" if not hyb else ("batch " + os.environ["VAMD_BATCH"] if os.environ.get("VAMD_BATCH") else "per-state"),
    N, sum(nb), wall, sum(nb) / wall, secs / wall, N * secs / wall))
