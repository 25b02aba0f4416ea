"""The 5.1 layout of libvorbisenc (reference lib/modes/setup_44p51.h, residue_44p51.h:283-290): six
channels on two submaps -- the five full-range channels share a floor and one interleaved type-2 residue,
the LFE has a floor and a type-1 residue of its own -- and four coupling steps in which the left channel
is the magnitude of three (lib/psy.c:1111-1201, "depth>1 coupling"); SURVEY.md 8f rank 4.

What pins what:
  * tests/golden/blocks_44k_51_q3.npz: blocks the reference's own blockout cut from a gated six-channel
    stream (q 0.3: coupling and noise normalisation both live), with its decisions (posts, quantised and
    coupled residue, partition classes, codebook entries) and the packets the real vorbis_analysis() wrote;
    the kernel bodies compiled for the host (CPU suite) and the HIP library (GPU suite) must reproduce them;
  * against the reference compiled in place: every quality region (coupled q < 0.5, uncoupled above; noise
    normalisation on below 0.4), silence, an LFE-only block, bitrate-managed candidates;
  * end to end: a 5.1 stream through the hybrid libvorbis -- block switching, all of mapping0_forward and the
    packets from the GPU -- emits the reference's bytes.
oracle/port (the sequential plain-C restatement) covers the layout too and is pinned the same way.
"""
import os

import numpy as np
import pytest

import vorbis_amd
from oracle import port, ref
from tests import checker, golden_io

ROOT = checker.ROOT
NAME = "44k_51_q3"
needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
KEYS = ("post_valid", "iwork", "nonzero", "local_ampmax")


def blob():
    return np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % NAME), dtype=np.uint8)


def block_args(b):
    return (b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])


def same_decisions(a, g, nposts):
    assert checker.compare_block(a, g, nposts, keys=KEYS, verbose=True) == 0
    assert np.array_equal(a["res_class"], g["res_class"])
    assert np.array_equal(a["res_entries"], g["res_entries"])


def surround_blocks(e, seed):
    """(pcm, lW, W, nW): plain noise, correlated fronts, silence, LFE only, one loud channel, short blocks."""
    rng = np.random.default_rng(seed)
    for it, amp in enumerate((0.5, 0.3, 0.0, 0.2, 0.9, 0.01, 0.6, 0.05)):
        W = 0 if it >= 6 else 1
        n = e.blocksize(W)
        pcm = ((rng.random((6, n), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        if it == 1:
            pcm[1] = 0.9 * pcm[0]
            pcm[4] = -0.5 * pcm[3]
        if it == 3:
            pcm[:5] = 0.0
        if it == 4:
            pcm[1:] *= 1e-4
        yield pcm, (it & 1 if W else 0), W, ((it >> 1) & 1 if W else 0)


# ------------------------------------------------------------------------------------------
# CPU suite
# ------------------------------------------------------------------------------------------
def test_kernel_bodies_match_golden():
    from tests.emul.emul import Emul
    blocks, posts, _ = golden_io.load(NAME)
    em, p = Emul(blob()), port.PortEncoder(blob())
    assert em.channels == 6 and em.L.emul_submaps(em.h, 1) == 2
    kinds = set()
    for b in blocks:
        same_decisions(b, p.tap_block(*block_args(b)), posts[b["W"]])
        g = em.analyze_block(*block_args(b))
        same_decisions(b, g, posts[b["W"]])
        assert np.array_equal(np.asarray(b["posts"])[:, :posts[b["W"]]], g["posts"][:, :posts[b["W"]]])
        assert g["packet"] == b["packet"]
        kinds.add((b["W"], b["blocktype"]))
    assert len(kinds) >= 3 and sum(len(b["res_entries"]) for b in blocks) > 3000


@needs_ref
@pytest.mark.parametrize("quality", [-0.1, 0.1, 0.3, 0.4, 0.6, 1.0])
def test_kernel_bodies_match_reference(quality):
    from tests.emul.emul import Emul
    e = ref.RefEncoder(6, 44100, quality)
    em, p = Emul(e.pack_setup()), port.PortEncoder(e.pack_setup())
    for pcm, lW, W, nW in surround_blocks(e, int(quality * 10) + 60):
        a = e.tap_block(pcm, lW, W, nW, 1 if W else 0)
        assert a["packet_matches_real"]
        o = p.tap_block(pcm, lW, W, nW, 1 if W else 0)
        assert checker.compare_block(a, o, e.floor_posts(W), verbose=True) == 0
        same_decisions(a, o, e.floor_posts(W))
        g = em.analyze_block(pcm, lW, W, nW, 1 if W else 0)
        assert checker.compare_block(a, g, e.floor_posts(W), verbose=True) == 0   # every float tap too
        same_decisions(a, g, e.floor_posts(W))
        assert g["packet"] == a["packet"] and len(a["packet"]) <= em.L.emul_packet_capacity(em.h, W)


@needs_ref
@pytest.mark.parametrize("rates", [(-1, 256000, -1), (-1, 160000, -1)])
def test_kernel_bodies_match_reference_managed(rates):
    from tests.emul.emul import Emul
    e = ref.RefEncoder(6, 44100, managed=rates)
    em, p = Emul(e.pack_setup()), port.PortEncoder(e.pack_setup())
    for pcm, lW, W, nW in list(surround_blocks(e, rates[1]))[:3] + list(surround_blocks(e, rates[1]))[6:7]:
        a = e.tap_block_managed(pcm, lW, W, nW, 1 if W else 0)
        assert a["packets_match_real"]
        o = p.tap_block_managed(pcm, lW, W, nW, 1 if W else 0)
        assert np.array_equal(a["m_iwork"], o["m_iwork"]) and np.array_equal(a["m_nonzero"], o["m_nonzero"])
        g = em.analyze_block_managed(pcm, lW, W, nW, 1 if W else 0)
        assert np.array_equal(a["m_iwork"], g["m_iwork"]) and np.array_equal(a["m_nonzero"], g["m_nonzero"])
        assert g["m_packets"] == a["m_packets"]


@needs_ref
def test_detector_kernel_bodies_match_reference():
    """The block-switching detector ORs its triggers over all six channels (lib/envelope.c:234-239)."""
    from tests.emul.emul import Emul
    from tests.test_envelope import gated, check_state
    from vorbis_amd import EnvelopeState, envelope_marks
    x = gated(6, 40000, 66)
    x[2] = np.roll(x[2], 2500)        # the centre channel's bursts fall between the others'
    e = ref.RefEncoder(6, 44100, 0.3)
    o = e.envelope_feed(x)
    assert o["marks"].sum() > 10
    st = EnvelopeState()
    flags = Emul(blob()).envelope_search(o["pcm"], o["steps"], st)
    assert np.array_equal(envelope_marks(flags)[:o["steps"] + 2], o["marks"])
    check_state(st, o, 6)


# ------------------------------------------------------------------------------------------
# GPU suite (through the C ABI)
# ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@needs_ref
def test_gpu_detector_matches_reference():
    from tests.test_envelope import gated, check_state
    from vorbis_amd import envelope_marks
    x = gated(6, 60000, 67)
    x[4] = np.roll(x[4], 1700)
    o = ref.RefEncoder(6, 44100, 0.3).envelope_feed(x)
    an = vorbis_amd.Analyzer(blob(), 0)
    flags, st = an.envelope_search(o["pcm"], o["steps"])
    assert np.array_equal(envelope_marks(flags)[:o["steps"] + 2], o["marks"]) and o["marks"].sum() > 10
    check_state(st, o, 6)


@pytest.mark.gpu
def test_gpu_matches_golden():
    import torch
    blocks, posts, _ = golden_io.load(NAME)
    an = vorbis_amd.Analyzer(blob(), 0)
    assert an.channels == 6 and an.submaps(1) == 2 and an.packet_capacity(1) > 0
    for b in blocks:     # per block, host memory
        g = an.analyze_block(*block_args(b))
        same_decisions(b, g, posts[b["W"]])
        pk, amp = an.encode_block(*block_args(b))
        assert pk[0] == b["packet"] and np.float32(amp) == np.float32(b["ampmax_out"])
    for W in (0, 1):     # batched, device memory
        bs = [b for b in blocks if b["W"] == W]
        pcm = torch.from_numpy(np.stack([b["pcm"] for b in bs])).cuda()
        o = an.analyze(pcm, W=W, lW=[b["lW"] for b in bs], nW=[b["nW"] for b in bs], blocktype=[b["blocktype"] for b in bs],
                       ampmax_in=[b["ampmax_in"] for b in bs],
                       want=("iwork", "nonzero", "res_class", "res_entries", "res_count", "packets", "packet_bits"))
        torch.cuda.synchronize()
        h = {k: v.cpu().numpy() for k, v in o.items()}
        for k, b in enumerate(bs):
            assert np.array_equal(h["iwork"][k], b["iwork"]) and np.array_equal(h["nonzero"][k], b["nonzero"])
            cls, ent = an.residue_lists(W, h["res_class"][k], h["res_entries"][k], h["res_count"][k])
            assert np.array_equal(cls, b["res_class"]) and np.array_equal(ent, b["res_entries"])
            assert vorbis_amd.packet_bytes(h["packets"][k], h["packet_bits"][k]) == b["packet"]


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("quality", [-0.1, 0.1, 0.6, 1.0])
def test_gpu_matches_reference(quality):
    e = ref.RefEncoder(6, 44100, quality)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    for pcm, lW, W, nW in surround_blocks(e, int(quality * 10) + 160):
        a = e.tap_block(pcm, lW, W, nW, 1 if W else 0)
        g = an.analyze_block(pcm, lW, W, nW, 1 if W else 0)
        same_decisions(a, g, e.floor_posts(W))
        assert np.array_equal(np.asarray(a["mdct"]).view(np.uint32), g["mdct"].view(np.uint32))
        pk, _ = an.encode_block(pcm, lW, W, nW, 1 if W else 0)
        assert pk[0] == a["packet"]


@pytest.mark.gpu
@needs_ref
def test_gpu_managed_packets():
    e = ref.RefEncoder(6, 44100, managed=(-1, 256000, -1))
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    for pcm, lW, W, nW in list(surround_blocks(e, 7))[:2] + list(surround_blocks(e, 7))[6:7]:
        a = e.tap_block_managed(pcm, lW, W, nW, 1 if W else 0)
        pk, _ = an.encode_block(pcm, lW, W, nW, 1 if W else 0, managed=True)
        assert pk == a["m_packets"]


@pytest.mark.gpu
@pytest.mark.skipif(not (ref.available() and ref.hybrid_available()), reason="oracle/_ref libraries not built")
@pytest.mark.parametrize("quality", [0.3, 0.7])
def test_hybrid_encode_emits_reference_packets(quality):
    rng = np.random.default_rng(51)
    frames = 44100
    t = np.arange(frames)
    x = (rng.random((6, frames), dtype=np.float32) - 0.5) * 2 * np.where((t % 11025) < 1102, 0.5, 0.0005)
    x[1] = 0.8 * x[0] + 0.2 * x[1]
    x[5] *= 0.05
    x = np.ascontiguousarray(x, dtype=np.float32)
    want = ref.RefEncoder(6, 44100, quality).encode_stream(x)
    got = ref.RefEncoder(6, 44100, quality, hybrid=True).encode_stream(x)
    assert len(want) == len(got) > 20
    assert [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in want] == [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in got]
    for k, (a, b) in enumerate(zip(want, got)):
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])


@pytest.mark.gpu
@needs_ref
def test_gpu_48k_surround_stream_api():
    """48 kHz 5.1 through the stream entry point (ampmax chained across blocks) straight to packets."""
    import torch
    e = ref.RefEncoder(6, 48000, 0.5)
    rng = np.random.default_rng(48)
    x = ((rng.random((6, 48000), dtype=np.float32) - 0.5) * 0.8).astype(np.float32)
    stream = e.encode_stream(x)
    run = stream[4:12]                                   # consecutive blocks of the stream (steady noise: all long)
    assert all(b["W"] == 1 and b["lW"] == 1 and b["nW"] == 1 for b in run)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    pcm = torch.from_numpy(np.stack([b["pcm"] for b in run])).cuda()
    outs, state = an.analyze_stream(pcm, stream[3]["ampmax_out"], W=1, want=("ampmax_out", "packets", "packet_bits"))
    torch.cuda.synchronize()
    rows, bits = outs["packets"].cpu().numpy(), outs["packet_bits"].cpu().numpy()
    for k, b in enumerate(run):
        assert vorbis_amd.packet_bytes(rows[k], bits[k]) == b["packet"], k
    assert np.float32(state) == np.float32(run[-1]["ampmax_out"])


# ---- the other channel counts Vorbis I assigns an order to (3, 4, 5, 6.1, 7.1): libvorbisenc sets them up
# uncoupled -- one submap, no coupling steps, a type-1 residue that codes every channel on its own ----------
@needs_ref
@pytest.mark.parametrize("ch", [3, 4, 5, 7, 8])
def test_other_channel_counts_port_and_kernel_bodies(ch):
    from tests.emul.emul import Emul
    rng = np.random.default_rng(ch)
    for q in (0.1, 0.6):
        e = ref.RefEncoder(ch, 44100, q)
        setup = e.pack_setup()
        em, p = Emul(setup), port.PortEncoder(setup)
        for W, amp in ((1, 0.5), (1, 0.0), (0, 0.3)):
            pcm = ((rng.random((ch, e.blocksize(W)), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
            if ch > 3:
                pcm[ch - 1] = 0.0          # a silent channel: one stream fewer in the residue
            a = e.tap_block(pcm, W, W, W, 1 if W else 0)
            assert a["packet_matches_real"]
            for g in (p.tap_block(pcm, W, W, W, 1 if W else 0), em.analyze_block(pcm, W, W, W, 1 if W else 0)):
                assert checker.compare_block(a, g, e.floor_posts(W), verbose=True) == 0
                same_decisions(a, g, e.floor_posts(W))
            assert g["packet"] == a["packet"]


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("ch", [3, 8])
def test_gpu_other_channel_counts(ch):
    rng = np.random.default_rng(10 + ch)
    e = ref.RefEncoder(ch, 44100, 0.4)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    for W, amp in ((1, 0.5), (0, 0.3), (1, 0.0)):
        pcm = ((rng.random((ch, e.blocksize(W)), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        pcm[ch - 1] *= 0.0
        a = e.tap_block(pcm, W, W, W, 1 if W else 0)
        g = an.analyze_block(pcm, W, W, W, 1 if W else 0)
        same_decisions(a, g, e.floor_posts(W))
        pk, _ = an.encode_block(pcm, W, W, W, 1 if W else 0)
        assert pk[0] == a["packet"]
