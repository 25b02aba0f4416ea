"""Regenerate tests/golden/envelope_*.npz from the reference itself: a short gated-noise stream is
written through vorbis_analysis_buffer/_wrote and the reference's own _ve_envelope_search decides the
marks (oracle/ref_harness.c ref_envelope_feed).  Stored: the PCM ring exactly as the detector saw it
(centre padding + start-of-stream pre-extrapolation included), the marks, and the final filter state.
Needs /root/reference (build container only); the fixtures travel with the repo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests.checker import SETUPS  # noqa: E402


def stream(ch, frames, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(frames)
    gate = np.where((t % 6000) < 900, 0.5, 0.002).astype(np.float32)
    return ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * gate).astype(np.float32)


if __name__ == "__main__":
    for name in ("44k_stereo_q4", "44k_mono_q5"):
        ch, rate, q = SETUPS[name]
        o = ref.RefEncoder(ch, rate, q).envelope_feed(stream(ch, 24000, 777))
        assert o["marks"].sum() > 8
        path = os.path.join(ROOT, "tests", "golden", "envelope_%s.npz" % name)
        np.savez_compressed(path, pcm=o["pcm"], marks=o["marks"], steps=np.array([o["steps"]]),
                            stretch=np.array([o["stretch"]]), near=o["near"], amp=o["amp"])
        print(name, o["steps"], "steps,", int(o["marks"].sum()), "marks,", os.path.getsize(path), "bytes")
