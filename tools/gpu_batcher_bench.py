"""Aggregate throughput of N encoder threads through libvorbis' own application loop, timed in C
(oracle/ref_harness.c: ref_time_threads -- one call, no Python inside the clock):

    python tools/gpu_batcher_bench.py N [stream seconds] [passes]         the hybrid libvorbis (GPU back-end)
    VAMD_BATCH=256 python tools/gpu_batcher_bench.py N ...                ... through the batcher
    VAMD_CPU_ONLY=1 python tools/gpu_batcher_bench.py N ...               the unmodified reference on the host CPUs

One mode per process (the binding reads VAMD_BATCH once).  Prints one line."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hyb = os.environ.get("VAMD_CPU_ONLY") is None
L = ref.lib(hybrid=hyb)
L.ref_time_threads.restype = C.c_double
L.ref_time_threads.argtypes = [C.c_int, C.c_int, C.c_long, C.c_float, C.POINTER(C.c_float), C.c_long, C.c_int,
                               C.POINTER(C.c_long), C.POINTER(C.c_double)]
rng = np.random.default_rng(1)
x = ((rng.random((2, int(44100 * secs)), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
xp = x.ctypes.data_as(C.POINTER(C.c_float))
if hyb:  # open the GPU side (context / batcher, code objects) before the clock starts, and keep it open
    keeper = ref.RefEncoder(2, 44100, 0.4, hybrid=True)
    keeper.encode_stream(x[:, :22050])
    warm_blocks, warm_cpu = C.c_long(0), (C.c_double * 2)()
    L.ref_time_threads(min(N, 16), 2, 44100, 0.4, xp, 22050, 1, C.byref(warm_blocks), warm_cpu)
blocks, cpu = C.c_long(0), (C.c_double * 2)()
wall = L.ref_time_threads(N, 2, 44100, 0.4, xp, x.shape[1], passes, C.byref(blocks), cpu)
assert wall > 0, "an encode failed"
mode = "cpu reference" if not hyb else ("batcher(%s)" % os.environ["VAMD_BATCH"] if os.environ.get("VAMD_BATCH") else "context per state")
line = "%-20s threads %4d  blocks %7d  wall %6.2f s  -> %8.0f blocks/s  host CPU %.3f ms user + %.3f ms system per block (%.1f CPUs busy)" % (
    mode, N, blocks.value, wall, blocks.value / wall, 1e3 * cpu[0] / blocks.value, 1e3 * cpu[1] / blocks.value, (cpu[0] + cpu[1]) / wall)
if hyb and os.environ.get("VAMD_BATCH"):
    a, b, t = C.c_long(0), C.c_long(0), C.c_double(0)
    L.vamd_batch_stats(C.byref(a), C.byref(b), C.byref(t))
    line += "  | batches %d, %.1f blocks each, %.3f ms inside the GPU call" % (a.value, b.value / max(a.value, 1), 1e3 * t.value / max(a.value, 1))
    if hasattr(L, "vamd_batch_trace"):
        buf = C.create_string_buffer(4096)
        L.vamd_batch_trace(buf, 4096)
        line += "\n    " + buf.value.decode().replace("\n", "\n    ")
print(line)
