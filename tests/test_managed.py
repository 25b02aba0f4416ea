"""Bitrate-managed blocks: the fifteen candidate packets mapping0_forward prepares per block when a
bitrate manager is active (reference lib/mapping0.c:507-573,596-687).

What pins what:
  * oracle/ref_harness.c restates the managed branch over the reference's extern functions and checks
    all fifteen candidate packets byte for byte against the real vorbis_analysis(vb, NULL);
  * oracle/port (sequential) and the kernel bodies compiled for the host must reproduce every candidate's
    posts, residue and flags -- CPU suite; the HIP library too, per block and batched -- GPU suite;
  * end to end: an ABR encode through the hybrid libvorbis (bitrate manager untouched, candidates from the
    GPU) emits the packets the CPU reference emits.
"""
import numpy as np
import pytest

import vorbis_amd
from oracle import port, ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
KEYS = ("m_post_valid", "m_iwork", "m_nonzero", "m_posts")


def blocks(e, seed):
    rng = np.random.default_rng(seed)
    for amp, W in ((0.5, 1), (0.01, 1), (1.0, 1), (0.0, 1), (0.7, 0), (0.003, 0)):
        n = e.blocksize(W)
        pcm = ((rng.random((e.channels, n), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        if amp == 1.0 and e.channels == 2:
            pcm[1] = pcm[0] * 0.6   # correlated channels: lossless coupling
        yield pcm, W


def same(a, b):
    for k in KEYS:
        x, y = a[k], b[k]
        if k == "m_posts":  # reference rows are 65 wide, ours 32; only `posts` entries are meaningful
            x, y = x[:, :, :29], y[:, :, :29]
        if not np.array_equal(x, y):
            return k
    return None


# (max, nominal, min) bitrates; 64 kb/s puts noise normalisation and the sliding lowpass inside the block
RATES = [(-1, 128000, -1), (-1, 64000, -1), (160000, 96000, 64000), (-1, 256000, -1)]


@pytest.mark.parametrize("rates", RATES)
def test_port_and_kernel_bodies_match_reference(rates):
    from tests.emul.emul import Emul
    e = ref.RefEncoder(2, 44100, managed=rates)
    blob = e.pack_setup()
    p, em = port.PortEncoder(blob), Emul(blob)
    for pcm, W in blocks(e, rates[1]):
        args = (pcm, W, W, W, 1 if W else 0)
        a = e.tap_block_managed(*args)
        assert a["packets_match_real"], "tap harness diverged from vorbis_analysis()"
        assert same(a, p.tap_block_managed(*args)) is None
        b = em.analyze_block_managed(*args)
        assert same(a, b) is None
        assert np.array_equal(a["mdct"].view(np.uint32), b["mdct"].view(np.uint32))
    # the fifteen candidates are not copies of one another
    assert len({bytes(x) for x in a["m_packets"]}) > 1


def test_mono_managed_port_matches_reference():
    e = ref.RefEncoder(1, 44100, managed=(-1, 64000, -1))
    p = port.PortEncoder(e.pack_setup())
    for pcm, W in blocks(e, 9):
        args = (pcm, W, W, W, 1 if W else 0)
        assert same(e.tap_block_managed(*args), p.tap_block_managed(*args)) is None


@pytest.mark.gpu
@pytest.mark.parametrize("rates", RATES[:3])
def test_gpu_matches_reference(rates):
    import torch
    e = ref.RefEncoder(2, 44100, managed=rates)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    for pcm, W in blocks(e, rates[1] + 1):
        args = (pcm, W, W, W, 1 if W else 0)
        a = e.tap_block_managed(*args)
        b = an.analyze_block_managed(*args)
        assert same(a, b) is None
        assert np.float32(a["ampmax_out"]) == np.float32(b["ampmax_out"])
    nb = 12
    rng = np.random.default_rng(3)
    pcm = ((rng.random((nb, 2, 2048), dtype=np.float32) - 0.5) * np.array([1.0, 0.02, 0.0])[np.arange(nb) % 3, None, None])
    pcm = pcm.astype(np.float32)
    o = an.analyze_managed(torch.from_numpy(pcm).cuda(), residue=True)
    torch.cuda.synchronize()
    for i in range(nb):
        a = e.tap_block_managed(pcm[i])
        got = {k: o[k][i].cpu().numpy() for k in KEYS}
        assert same(a, got) is None, i
    cnt = o["m_res_count"].cpu().numpy()
    assert (cnt[:, :, 1] <= an.residue_capacity(1)).all() and cnt[0, :, 1].min() > 0 and cnt[2, :, 1].max() == 0


@pytest.mark.gpu
@pytest.mark.skipif(not ref.hybrid_available(), reason="hybrid library not built")
@pytest.mark.parametrize("ch,rates,write", [(2, (-1, 96000, -1), 1024), (2, (-1, 64000, -1), 1024), (1, (-1, 64000, -1), 1024),
                                            (2, (-1, 96000, -1), 30000), (2, (128000, 96000, 64000), 65536)])
def test_hybrid_abr_encode_emits_reference_packets(ch, rates, write):
    """(write > 1024: the binding's look-ahead -- the buffered blocks' fifteen candidate packets each come out of ONE batch,
    vamd_encode_blocks(managed), and the bitrate manager picks among them block by block as before.)"""
    rng = np.random.default_rng(77)
    frames = 44100 * (2 if write == 1024 else 5)
    t = np.arange(frames)
    x = ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * np.where((t % 11025) < 1102, 0.5, 0.0005))
    x = np.ascontiguousarray(x, dtype=np.float32)
    want = ref.RefEncoder(ch, 44100, managed=rates).encode_stream(x, write_frames=write)
    got = ref.RefEncoder(ch, 44100, managed=rates, hybrid=True).encode_stream(x, write_frames=write)
    assert len(want) == len(got) > 40
    assert [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in want] == \
           [(b["lW"], b["W"], b["nW"], b["blocktype"]) for b in got]
    for k, (a, b) in enumerate(zip(want, got)):
        assert a["packet"] == b["packet"], "packet %d differs (W=%d)" % (k, a["W"])
    assert sum(1 for b in want if b["W"] == 0) > 10
