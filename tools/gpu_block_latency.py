"""Scratch: latency of the per-block host entry points (the compatibility path behind vorbis_analysis())."""
import os, sys, time
os.environ["VAMD_TEST_KNOBS"] = "1"  # (VAMD_NO_OVERLAP is a test knob: vorbis_amd/csrc/vamd_knobs.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vorbis_amd
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
rng = np.random.default_rng(0)
pcm = ((rng.random((2, 2048), dtype=np.float32) - 0.5)).astype(np.float32)
for name, fn in (("encode_block (PCM -> packet)", lambda: an.encode_block(pcm)),
                 ("analyze_block (tensors back)", lambda: an.analyze_block(pcm)),
                 ("envelope_search (16 steps)", lambda: an.envelope_search(np.zeros((2, 15 * 64 + 128), np.float32) + pcm[:, :15 * 64 + 128], 16))):
    for _ in range(20):
        fn()
    t0 = time.time()
    N = 300
    for _ in range(N):
        fn()
    print("%-32s %.0f us per call" % (name, (time.time() - t0) / N * 1e6))
for env in ("VAMD_NO_OVERLAP",):
    os.environ[env] = "1"
    an2 = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
    for _ in range(20):
        an2.encode_block(pcm)
    t0 = time.time()
    for _ in range(300):
        an2.encode_block(pcm)
    print("encode_block with %s: %.0f us per call" % (env, (time.time() - t0) / 300 * 1e6))
