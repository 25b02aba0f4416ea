"""CPU suite: the whole-stream sequence (vamd_plan_streams_whole) out of the product's own kernel bodies compiled for the host."""
import numpy as np
import pytest


@pytest.mark.parametrize("frames,kind", [(30000, "gated"), (9000, "noise"), (2500, "gated"), (700, "noise"), (3072, "sine"), (20, "noise")])
def test_whole_stream_sequence_on_the_host_matches_the_reference(frames, kind):
    """vamd_plan_streams_whole's pieces -- the two LPC stream ends (k_lpc.h), the detector, the walk with the end of the
    stream (k_blockout.h: eof) -- compiled for the host (tests/emul) and chained as the library chains them, against the
    reference's application loop: the block list to the last (e_o_s) block and every block's SAMPLES, the extrapolated
    ones bit for bit.  (No GPU: the systolic predictor and the lag-per-lane sums of the GPU build are held against the
    reference by tests/test_feed.py.)"""
    from oracle import ref
    from tests.emul.emul import Emul
    import vorbis_amd
    if not ref.available():
        pytest.skip("needs the reference build")
    rng = np.random.default_rng(frames)
    t = np.arange(frames)
    if kind == "gated":
        x = (rng.random((2, frames), dtype=np.float32) - 0.5) * 2 * np.where((t % 6000) < 500, 0.5, 0.0005)
    elif kind == "sine":
        x = 0.5 * np.sin(0.05 * t)[None, :] * np.ones((2, 1)) + (rng.random((2, frames)) - 0.5) * 1e-3
    else:
        x = (rng.random((2, frames), dtype=np.float32) - 0.5)
    x = np.ascontiguousarray(x, dtype=np.float32)
    want = ref.RefEncoder(2, 44100, 0.4).encode_stream(x)
    em = Emul(vorbis_amd.default_setup_blob("44k_stereo_q4"))
    buf, kinds, begins = em.whole_stream(x)
    assert len(kinds) == len(want), (len(kinds), len(want))
    for k, b in enumerate(want):
        W = int(kinds[k]) & 1
        assert (W, (int(kinds[k]) >> 1) & 1, (int(kinds[k]) >> 2) & 1, (int(kinds[k]) >> 3) & 1) == (b["W"], b["lW"], b["nW"], b["blocktype"]), k
        n = em.bs[W]
        got = buf[:, int(begins[k]):int(begins[k]) + n]
        assert np.array_equal(got.view(np.uint32), b["pcm"].view(np.uint32)), "block %d of %d: samples differ" % (k, len(want))
    assert want[-1]["eos"] == 1
