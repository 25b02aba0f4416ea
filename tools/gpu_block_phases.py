"""Scratch: the in-kernel phase stopwatch over the per-block entry point (one stereo block per call): where a lone
wave's microseconds go, kernel by kernel.  Rows: transform, noise, tone (seed / chase / fold slots), floor, couple +
residue (slots 8, 9) + pack (slots 10..15: floor values, offsets, stage 0, later stages, tail, floor fields)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vorbis_amd
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
rng = np.random.default_rng(0)
pcm = ((rng.random((2, 2048), dtype=np.float32) - 0.5)).astype(np.float32)
for _ in range(20):
    an.encode_block(pcm)
an.debug_cycles(True)
N = 200
for _ in range(N):
    an.encode_block(pcm)
c = an.debug_cycles(False, read=True)
# clock64 (s_memtime) counts shader-clock ticks on gfx950 (2.4 GHz measured over such runs: bench.py's shader_clock); a
# slot sums the ticks of ALL the stage's waves (a lone stereo block: 2 for the floor, the kernels' teams elsewhere)
GHZ = 2.4
for name, row in zip(("transform", "noise", "tone", "floor", "couple/residue/pack"), c):
    print("%-20s us per call by slot, summed over the stage's waves: %s" % (name, [round(float(x) / N / (GHZ * 1e3), 1) for x in row]))
