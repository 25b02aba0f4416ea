"""GPU suite: the drop-in boundary end to end.

oracle/_ref/libvorbis_hybrid.so is the reference's own libvorbis objects with lib/mapping0.c replaced
by the binding integration/mapping0_vamd.c (all of mapping0_forward -> one call into libvorbis_amd.so
that returns the block's packet bytes; for modes whose residue the GPU does not search, the numeric
section only and the reference's bit-writing half unchanged) and lib/envelope.c replaced by
integration/envelope_vamd.c (the block-switching detector's steps -> libvorbis_amd.so, mark/cursor
bookkeeping unchanged), so on the gated streams below the GPU also decides where the short blocks go.
Driving the unmodified application loop (vorbis_analysis_buffer / _wrote / _blockout / vorbis_analysis,
examples/encoder_example.c:179-236) through it must yield the very packets the pure CPU reference
emits -- the strongest parity statement the domain offers: every bit of every packet was produced on
the GPU and is identical to what the reference's own analysis, Huffman and VQ code writes.
"""
import numpy as np
import pytest

from oracle import ref

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ref.available() and ref.hybrid_available()),
                                 reason="oracle/_ref libraries not built (needs /root/reference at build time)")]


def _stream(ch, seconds, kind, seed):
    rng = np.random.default_rng(seed)
    frames = int(44100 * seconds)
    x = (rng.random((ch, frames), dtype=np.float32) - 0.5)
    if kind == "gated":   # transients -> short blocks, transitions (SURVEY.md 8d "C5")
        t = np.arange(frames)
        x *= 2 * np.where((t % 11025) < 1102, 0.5, 0.0005).astype(np.float32)
    elif kind == "s16":   # encoder_example.c:197-202: s16 samples / 32768
        x = np.round(x * 32767).astype(np.int16).astype(np.float32) / 32768.0
    return np.ascontiguousarray(x, dtype=np.float32)


@pytest.mark.parametrize("ch,q,kind", [(2, 0.4, "s16"), (2, 0.9, "gated"), (2, 0.1, "gated"), (1, 0.5, "gated")])
def test_hybrid_encode_emits_reference_packets(ch, q, kind):
    assert hasattr(ref.lib(hybrid=True), "_ve_envelope_search_cpu")  # the detector binding is linked in
    assert hasattr(ref.lib(hybrid=True), "vamd_encode_block")        # ... and packets come from the GPU library
    pcm = _stream(ch, 1.0 if kind == "s16" else 2.0, kind, seed=12345)
    want = ref.RefEncoder(ch, 44100, q).encode_stream(pcm)
    got = ref.RefEncoder(ch, 44100, q, hybrid=True).encode_stream(pcm)
    assert len(want) == len(got) > 40
    assert [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in want] == \
           [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in got]
    for k, (a, b) in enumerate(zip(want, got)):
        assert np.float32(a["ampmax_in"]) == np.float32(b["ampmax_in"]), k     # the ampmax chain stays in step
        assert np.float32(a["ampmax_out"]) == np.float32(b["ampmax_out"]), k
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])
    if kind == "gated":
        assert sum(1 for b in want if b["W"] == 0) > 10
