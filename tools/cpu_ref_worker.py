"""One process of bench.py's cpu_baseline process-per-core leg: one RefEncoder (oracle/_ref: the unmodified reference) or, when
that library is absent, one PortEncoder (oracle/port), timing the DSP part of mapping0_forward over the same seeded white-noise
blocks from `start` (a time.time() value every worker is given) for `seconds`.  Imports numpy and ctypes only -- no torch -- so
that a hundred of them start in a second.  Prints: blocks done, wall seconds, CPU seconds of this process.

    python tools/cpu_ref_worker.py <setup> <start> <seconds>
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

setup, start, seconds = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
SETUPS = {"44k_stereo_q4": (2, 44100, 0.4), "44k_stereo_q9": (2, 44100, 0.9), "44k_stereo_q1": (2, 44100, 0.1),
          "44k_mono_q5": (1, 44100, 0.5), "44k_51_q3": (6, 44100, 0.3)}
ch, rate, q = SETUPS[setup]
from oracle import ref  # noqa: E402
if ref.available():
    enc = ref.RefEncoder(ch, rate, q)
else:
    from oracle import port
    enc = port.PortEncoder(np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % setup), dtype=np.uint8))
sample = 64
pcm = (np.random.default_rng(99).random((sample, ch, 2048), dtype=np.float32) - 0.5).astype(np.float32)
enc.time_dsp(pcm[:4], 1)  # warm
while time.time() < start:
    time.sleep(0.001)
t0, c0, done = time.time(), time.process_time(), 0
while time.time() < start + seconds:
    enc.time_dsp(pcm, 1)
    done += sample
print(done, time.time() - t0, time.process_time() - c0)
