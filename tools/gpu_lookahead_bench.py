"""Run on the GPU box: ONE application thread, one stream, through the hybrid libvorbis (GPU back-end, look-ahead inside
the stream) and through the unmodified reference, for several sizes of vorbis_analysis_wrote() -- timed in C
(oracle/ref_harness.c: ref_time_threads_w, one call, no Python inside the clock).  VERDICT r04 missing 3 / next 5.

    python tools/gpu_lookahead_bench.py [seconds of audio] [q]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import ref  # noqa: E402
from tests.test_gpu_dropin import _stream  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
q = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
kind = sys.argv[3] if len(sys.argv) > 3 else "s16"
nominal = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # > 0: bitrate-managed encoders (ABR at that rate) instead of VBR at q
x = _stream(2, secs, kind, seed=77)
xp = x.ctypes.data_as(C.POINTER(C.c_float))


def timed(hybrid, write_frames, passes=1):
    L = ref.lib(hybrid)
    L.ref_time_set_managed(C.c_long(nominal))
    L.ref_time_threads_w.restype = C.c_double
    L.ref_time_threads_w.argtypes = [C.c_int, C.c_int, C.c_long, C.c_float, C.POINTER(C.c_float), C.c_long, C.c_long, C.c_int,
                                     C.POINTER(C.c_long), C.POINTER(C.c_double)]
    blocks, cpu = C.c_long(0), (C.c_double * 2)()
    L.ref_time_threads_w(1, 2, 44100, q, xp, min(x.shape[1], 44100), write_frames, 1, C.byref(blocks), cpu)   # warm-up
    wall = L.ref_time_threads_w(1, 2, 44100, q, xp, x.shape[1], write_frames, passes, C.byref(blocks), cpu)
    return blocks.value, wall, cpu[0] + cpu[1]


print("# one thread, one %s stereo stream of %.0f s, %s; blocks/s wall (and per host-CPU-second)"
      % (kind, secs, ("bitrate-managed, %d bit/s nominal (fifteen candidate packets per block)" % nominal) if nominal else "q %.1f" % q))
print("# %10s %22s %22s %s" % ("write size", "reference (CPU)", "hybrid (GPU back-end)", "look-ahead hits / misses / batches"))
Lh = ref.lib(True)
for wf in (1024, 4096, 8192, 16384, 32768, 65536, 131072):
    nb_r, wall_r, cpu_r = timed(False, wf)
    h = [C.c_long(0) for _ in range(3)]
    Lh.vamd_ahead_stats(*[C.byref(v) for v in h])
    before = [v.value for v in h]
    nb_h, wall_h, cpu_h = timed(True, wf)
    Lh.vamd_ahead_stats(*[C.byref(v) for v in h])
    d = [v.value - b for v, b in zip(h, before)]
    print("%12d %10.0f (%8.0f) %12.0f (%8.0f)   %d / %d / %d   [%d vs %d blocks]" % (
        wf, nb_r / wall_r, nb_r / max(cpu_r, 1e-9), nb_h / wall_h, nb_h / max(cpu_h, 1e-9), d[0], d[1], d[2], nb_r, nb_h))
os.environ["VAMD_LOOKAHEAD"] = "0"
