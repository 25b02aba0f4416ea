"""Packet assembly on the device: the bit-writing half of mapping0_forward (reference
lib/mapping0.c:598-606 header bits, lib/floor1.c:833-921 floor1_encode's writes, lib/res0.c:534-640
_01forward's phrase words and codewords, lib/codebook.c:146-151, oggpack_write); SURVEY.md 8f rank 4.

What pins what:
  * tests/golden/blocks_*.npz carry, for every fixture block, the packet bytes the reference's real
    vorbis_analysis() emitted (tools/make_golden.py): the kernel bodies compiled for the host -- CPU
    suite -- and the HIP library -- GPU suite, per block and batched -- must reproduce them byte for byte;
  * against the reference compiled in place: every libvorbisenc quality, all four lW/nW flag
    combinations, silence, bitrate-managed blocks (all fifteen candidate packets);
  * the capacity bound (vamd_packet_capacity) must hold and a row that is too short must say so.
"""
import os

import numpy as np
import pytest

import vorbis_amd
from oracle import ref
from tests import checker, golden_io

ROOT = checker.ROOT
NAMES = list(checker.SETUPS)
needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def blob_of(name):
    return np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % name), dtype=np.uint8)


def block_args(b):
    return (b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])


def random_blocks(e, seed):
    """(pcm, lW, W, nW): loud / quiet / full-scale / silent / spiky long blocks with every flag pair, two short."""
    rng = np.random.default_rng(seed)
    for it, amp in enumerate((0.5, 0.01, 1.0, 0.0, 0.9, 1e-4, 0.7, 0.02)):
        W = 0 if it >= 6 else 1
        n = e.blocksize(W)
        pcm = ((rng.random((e.channels, n), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        if it == 4:
            pcm[:, ::7] *= -1.0
        yield pcm, (it & 1 if W else 0), W, ((it >> 1) & 1 if W else 0)


# ------------------------------------------------------------------------------------------
# CPU suite
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", NAMES)
def test_kernel_bodies_match_golden_packets(name):
    from tests.emul.emul import Emul
    blocks, _, _ = golden_io.load(name)
    em = Emul(blob_of(name))
    sizes = []
    for b in blocks:
        g = em.analyze_block(*block_args(b))
        assert g["packet"] == b["packet"], (b["W"], b["blocktype"])
        assert (g["packet_bits"] + 7) // 8 == len(b["packet"])
        assert len(b["packet"]) <= em.L.emul_packet_capacity(em.h, b["W"])
        sizes.append(len(b["packet"]))
    assert min(sizes) == 1 and max(sizes) > 100   # silence (header + "no floor" flags only) and real packets


@needs_ref
@pytest.mark.parametrize("ch,quality", [(2, -0.1), (2, 0.1), (2, 0.4), (2, 0.7), (2, 0.9), (2, 1.0), (1, 0.0), (1, 0.5), (1, 1.0)])
def test_kernel_bodies_match_reference(ch, quality):
    from tests.emul.emul import Emul
    e = ref.RefEncoder(ch, 44100, quality)
    em = Emul(e.pack_setup())
    seen = 0
    for pcm, lW, W, nW in random_blocks(e, int(quality * 10) + ch):
        cap = em.L.emul_packet_capacity(em.h, W)
        assert cap > 0 and cap % 4 == 0
        a = e.tap_block(pcm, lW, W, nW, 1 if W else 0)
        assert a["packet_matches_real"]
        g = em.analyze_block(pcm, lW, W, nW, 1 if W else 0)
        assert g["packet"] == a["packet"], (W, lW, nW)
        assert len(a["packet"]) <= cap
        seen += 1
    assert seen >= 2


@needs_ref
@pytest.mark.parametrize("ch,rates", [(2, (-1, 128000, -1)), (2, (-1, 64000, -1)), (2, (160000, 96000, 64000)), (1, (-1, 48000, -1))])
def test_kernel_bodies_match_reference_managed(ch, rates):
    from tests.emul.emul import Emul
    e = ref.RefEncoder(ch, 44100, managed=rates)
    em = Emul(e.pack_setup())
    for pcm, lW, W, nW in random_blocks(e, rates[1]):
        a = e.tap_block_managed(pcm, lW, W, nW, 1 if W else 0)
        assert a["packets_match_real"]
        g = em.analyze_block_managed(pcm, lW, W, nW, 1 if W else 0)
        for k in range(15):
            assert g["m_packets"][k] == a["m_packets"][k], (W, k)


@needs_ref
def test_capacity_of_every_libvorbisenc_layout():
    """Every layout libvorbisenc sets up at 44.1 kHz has its packets assembled on the device: uncoupled stereo
    (a type-1 residue over two channels), the 4096-sample blocks of q < 0 (128 partitions), 5.1, mono."""
    from tests.emul.emul import Emul
    for ch, q, coupled in ((2, 0.4, False), (2, -0.1, True), (6, 0.4, True), (6, 0.9, True), (1, 0.3, True)):
        em = Emul(ref.RefEncoder(ch, 44100, q, coupled=coupled).pack_setup())
        assert em.L.emul_packet_capacity(em.h, 0) > 0 and em.L.emul_packet_capacity(em.h, 1) > 0, (ch, q, coupled)


# ------------------------------------------------------------------------------------------
# GPU suite (through the C ABI)
# ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_encode_block_matches_golden(name):
    blocks, _, _ = golden_io.load(name)
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    for b in blocks:
        pk, amp = an.encode_block(*block_args(b))
        assert pk[0] == b["packet"], (b["W"], b["blocktype"])
        assert np.float32(amp) == np.float32(b["ampmax_out"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_batch_matches_golden(name):
    """The fixture blocks of one size class as a batch with per-block lW / nW / blocktype / ampmax arrays."""
    import torch
    blocks, _, _ = golden_io.load(name)
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    for W in (0, 1):
        bs = [b for b in blocks if b["W"] == W]
        if not bs:
            continue
        pcm = torch.from_numpy(np.stack([b["pcm"] for b in bs])).cuda()
        o = an.analyze(pcm, W=W, lW=[b["lW"] for b in bs], nW=[b["nW"] for b in bs],
                       blocktype=[b["blocktype"] for b in bs], ampmax_in=[b["ampmax_in"] for b in bs],
                       want=("packets", "packet_bits"))
        torch.cuda.synchronize()
        rows, bits = o["packets"].cpu().numpy(), o["packet_bits"].cpu().numpy()
        for k, b in enumerate(bs):
            assert vorbis_amd.packet_bytes(rows[k], bits[k]) == b["packet"], (W, k)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("ch,quality", [(2, 0.1), (2, 0.4), (2, 1.0), (1, 0.5)])
def test_gpu_batch_matches_reference(ch, quality):
    """Larger random batches (packets well past the kernel's LDS ring), residue outputs asked for or not."""
    import torch
    e = ref.RefEncoder(ch, 44100, quality)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    rng = np.random.default_rng(17)
    for W, nb in ((1, 40), (0, 24)):
        n = an.blocksizes[W]
        amp = np.array([0.5, 0.01, 1.0, 0.0, 0.9])[np.arange(nb) % 5, None, None]
        pcm = ((rng.random((nb, ch, n), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        want = ("packets", "packet_bits") + (("res_class", "res_entries", "res_count") if W else ())
        o = an.analyze(torch.from_numpy(pcm).cuda(), W=W, lW=W, nW=0, blocktype=1 if W else 0, want=want)
        torch.cuda.synchronize()
        rows, bits = o["packets"].cpu().numpy(), o["packet_bits"].cpu().numpy()
        for k in range(nb):
            a = e.tap_block(pcm[k], W, W, 0, 1 if W else 0)
            assert vorbis_amd.packet_bytes(rows[k], bits[k]) == a["packet"], (W, k)
        assert not W or bits.max() > 8 * 256 * 4 or quality < 0.9   # (the ring holds 1 KB: long q10 packets wrap it)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("rates", [(-1, 128000, -1), (-1, 64000, -1)])
def test_gpu_managed_packets(rates):
    import torch
    e = ref.RefEncoder(2, 44100, managed=rates)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    blocks = list(random_blocks(e, 5))
    for pcm, lW, W, nW in blocks:     # per block, host memory (the binding's call)
        a = e.tap_block_managed(pcm, lW, W, nW, 1 if W else 0)
        pk, amp = an.encode_block(pcm, lW, W, nW, 1 if W else 0, managed=True)
        assert pk == a["m_packets"], W
    longs = [b for b in blocks if b[2] == 1]   # batched
    o = an.analyze_managed(torch.from_numpy(np.stack([b[0] for b in longs])).cuda(), W=1, lW=[b[1] for b in longs],
                           nW=[b[3] for b in longs], packets=True)
    torch.cuda.synchronize()
    rows, bits = o["m_packets"].cpu().numpy(), o["m_packet_bits"].cpu().numpy()
    for k, (pcm, lW, W, nW) in enumerate(longs):
        a = e.tap_block_managed(pcm, lW, W, nW, 1)
        assert [vorbis_amd.packet_bytes(rows[k, j], bits[k, j]) for j in range(15)] == a["m_packets"], k


@pytest.mark.gpu
def test_gpu_short_rows_and_argument_errors():
    import torch
    an = vorbis_amd.Analyzer(blob_of("44k_stereo_q9"), 0)
    cap = an.packet_capacity(1)
    assert cap > 0 and cap % 4 == 0 and an.packet_capacity(0) > 0 and an.L.vamd_packet_capacity(None, 1) == 0
    rng = np.random.default_rng(3)
    pcm = torch.from_numpy(((rng.random((3, 2, 2048), dtype=np.float32) - 0.5) * 1.6).astype(np.float32)).cuda()
    full = an.analyze(pcm, want=("packets", "packet_bits"))
    # a row shorter than the packet: the true length is still reported, the row holds the packet's head,
    # and nothing is written past the row
    outs = {"packets": torch.full((3, 256), 0x5A, dtype=torch.uint8, device="cuda"),
            "packet_bits": torch.zeros(3, dtype=torch.int32, device="cuda")}
    short = an.analyze(pcm[:, :, :], outs={"packets": outs["packets"][:, :128].contiguous(), "packet_bits": outs["packet_bits"]})
    torch.cuda.synchronize()
    assert torch.equal(short["packet_bits"], full["packet_bits"]) and int(full["packet_bits"].min()) > 8 * 128
    assert torch.equal(short["packets"], full["packets"][:, :128])
    with pytest.raises(ValueError):
        vorbis_amd.packet_bytes(short["packets"][0].cpu().numpy(), int(short["packet_bits"][0]))
    # argument errors
    with pytest.raises(vorbis_amd.VamdError) as ei:      # the two outputs go together
        an.analyze(pcm, outs={"packet_bits": outs["packet_bits"]})
    assert ei.value.code == -131
    with pytest.raises(vorbis_amd.VamdError) as ei:      # level
        an.analyze(pcm, level=vorbis_amd.LEVEL_PSY, outs=an.alloc_outputs(1, 3, ("packets", "packet_bits")))
    assert ei.value.code == -131
    bad = an.alloc_outputs(1, 3, ("packets", "packet_bits"))
    bad["packets"] = torch.zeros((3, 130), dtype=torch.uint8, device="cuda")   # stride not a multiple of 4
    with pytest.raises(vorbis_amd.VamdError) as ei:
        an.analyze(pcm, outs=bad)
    assert ei.value.code == -131


@pytest.mark.gpu
@needs_ref
def test_gpu_large_batch_takes_the_one_wave_kernels():
    """Past 2048 units the stereo residue is searched out of registers by a wave per unit (k_residue_chunks: runs of
    eight values, partitions of 32 values on long blocks and of 16 on short ones) and the packets are assembled by
    persistent waves (k_pack_waves); below, by four waves a unit through LDS (k_residue) and by two waves a packet
    (k_pack_pair).  The same blocks both ways give the same rows, and a sample of them is the reference's."""
    import torch
    e = ref.RefEncoder(2, 44100, 0.4)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    rng = np.random.default_rng(99)
    nb = 2304
    for W in (1, 0):
        amp = np.array([0.5, 0.01, 1.0, 0.0, 0.9, 0.2, 0.05])[np.arange(nb) % 7, None, None]
        pcm = ((rng.random((nb, 2, an.blocksizes[W]), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        dev = torch.from_numpy(pcm).cuda()
        big = an.analyze(dev, W=W, lW=W, nW=W, blocktype=1 if W else 0, want=("packets", "packet_bits"))
        torch.cuda.synchronize()
        rows, bits = big["packets"].cpu().numpy(), big["packet_bits"].cpu().numpy()
        for lo in range(0, nb, 768):
            part = an.analyze(dev[lo:lo + 768], W=W, lW=W, nW=W, blocktype=1 if W else 0, want=("packets", "packet_bits"))
            torch.cuda.synchronize()
            prow, pbits = part["packets"].cpu().numpy(), part["packet_bits"].cpu().numpy()
            assert np.array_equal(pbits, bits[lo:lo + 768])
            for k in range(768):
                assert vorbis_amd.packet_bytes(prow[k], pbits[k]) == vorbis_amd.packet_bytes(rows[lo + k], bits[lo + k]), (W, lo + k)
        for k in rng.choice(nb, 24, replace=False):
            a = e.tap_block(pcm[k], W, W, W, 1 if W else 0)
            assert vorbis_amd.packet_bytes(rows[k], bits[k]) == a["packet"], (W, k)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("rate,quality", [(8000, 0.3), (22050, 0.5), (32000, 0.1), (44100, 1.0), (48000, 0.7), (96000, 0.5)])
def test_gpu_batch_kernels_agree_across_rate_families(rate, quality):
    """The batch kernels of round 6 (k_residue_chunks: a run of eight values per lane; k_pack_waves: persistent waves) against
    the small-batch ones (k_residue: the work vector in LDS; k_pack_pair) on the same random blocks, both size classes, in
    every libvorbisenc rate family -- other partition sizes, other books, other floors than the 44.1 kHz fixtures -- and a
    sample of the rows against the reference."""
    import torch
    e = ref.RefEncoder(2, rate, quality)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    rng = np.random.default_rng(rate)
    nb = 2560
    for W in ((1, 0) if an.blocksizes[0] != an.blocksizes[1] else (0,)):  # (8 kHz: one block size, every block is W = 0)
        amp = np.array([0.5, 0.003, 1.0, 0.0, 0.9, 0.1, 0.02, 0.3])[rng.integers(0, 8, nb), None, None]
        pcm = ((rng.random((nb, 2, an.blocksizes[W]), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        dev = torch.from_numpy(pcm).cuda()
        big = an.analyze(dev, W=W, lW=W, nW=W, blocktype=1 if W else 0, want=("packets", "packet_bits", "res_class", "res_entries", "res_count"))
        torch.cuda.synchronize()
        rows, bits = big["packets"].cpu().numpy(), big["packet_bits"].cpu().numpy()
        cnt, ent = big["res_count"].cpu().numpy(), big["res_entries"].cpu().numpy()
        for lo in range(0, nb, 640):
            part = an.analyze(dev[lo:lo + 640], W=W, lW=W, nW=W, blocktype=1 if W else 0,
                              want=("packets", "packet_bits", "res_class", "res_entries", "res_count"))
            torch.cuda.synchronize()
            assert np.array_equal(part["packet_bits"].cpu().numpy(), bits[lo:lo + 640]), (W, lo)
            assert np.array_equal(part["res_count"].cpu().numpy(), cnt[lo:lo + 640]), (W, lo)
            pr, pe = part["packets"].cpu().numpy(), part["res_entries"].cpu().numpy()
            for k in range(640):
                assert vorbis_amd.packet_bytes(pr[k], bits[lo + k]) == vorbis_amd.packet_bytes(rows[lo + k], bits[lo + k]), (W, lo + k)
                n = int(cnt[lo + k].reshape(-1, 2)[:, 1].max())
                assert np.array_equal(pe[k][:n], ent[lo + k][:n]), (W, lo + k)
        for k in rng.choice(nb, 6, replace=False):
            a = e.tap_block(pcm[k], W, W, W, 1 if W else 0)
            assert vorbis_amd.packet_bytes(rows[k], bits[k]) == a["packet"], (W, k)


@pytest.mark.gpu
@needs_ref
def test_gpu_many_streams_in_one_call():
    """An encoder farm's batch: four real streams (block decisions taken by the reference's own blockout, short and
    long blocks mixed, each with its own ampmax chain) analysed by ONE vamd_analyze_streams_mixed call, straight
    to packets; every packet of every stream must be the reference's."""
    rng = np.random.default_rng(2024)
    e = ref.RefEncoder(2, 44100, 0.5)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    streams = []
    for s in range(4):
        frames = 30000 + 9000 * s
        t = np.arange(frames)
        gate = np.where((t % (7000 + 900 * s)) < 800, 0.5, 0.001 * (s + 1)).astype(np.float32)
        x = ((rng.random((2, frames), dtype=np.float32) - 0.5) * 2 * gate).astype(np.float32)
        streams.append(ref.RefEncoder(2, 44100, 0.5).encode_stream(x))
    assert all(any(b["W"] == 0 for b in st) and any(b["W"] == 1 for b in st) for st in streams)
    # every stream starts from a fresh psy_g_look.ampmax (-9999); its chain then runs through its own blocks
    res, states = an.analyze_streams_mixed(streams, [-9999.0] * len(streams), want=("ampmax_out", "packets", "packet_bits"))
    for s, (st, out) in enumerate(zip(streams, res)):
        assert len(st) == len(out) > 10
        for k, (b, o) in enumerate(zip(st, out)):
            assert vorbis_amd.packet_bytes(o["packets"], o["packet_bits"]) == b["packet"], (s, k, b["W"])
        assert np.float32(states[s]) == np.float32(st[-1]["ampmax_out"])


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("ch,quality,managed", [(2, 0.9, None), (2, 0.4, None), (1, 0.5, None), (2, None, (-1, 128000, -1))])
def test_gpu_encode_blocks_is_the_reference_stream(ch, quality, managed):
    """vamd_encode_blocks through the C ABI: stretches of CONSECUTIVE blocks of a real stream (short, long and transition
    blocks as the reference's blockout cut them), the first block's incoming ampmax given and the chain between the blocks
    on the device.  VBR: packets and both ends of every block's ampmax equal the reference's own stream.  Managed: the
    fifteen candidates of every block equal vamd_encode_block's, block by block along the same chain."""
    kw = dict(managed=managed) if managed else {}
    args = (ch, 44100) if managed else (ch, 44100, quality)
    e = ref.RefEncoder(*args, **kw)
    an = vorbis_amd.Analyzer(e.pack_setup(), 0)
    rng = np.random.default_rng(77)
    frames = 44100 * 3
    t = np.arange(frames)
    x = (rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * np.where((t % 9000) < 900, 0.5, 0.0005).astype(np.float32)
    blocks = ref.RefEncoder(*args, **kw).encode_stream(np.ascontiguousarray(x, dtype=np.float32))
    assert len(blocks) > 120 and {b["W"] for b in blocks} == {0, 1}
    for lo, hi in ((0, 1), (1, 3), (3, 70), (70, 71), (71, len(blocks))):   # one block, a pair, 67, one, the rest
        part = blocks[lo:hi]
        if managed and len(part) > 63:
            part = part[:63]
        pk, ain, aout, verdict = an.encode_blocks([b["pcm"] for b in part], [b["lW"] for b in part], [b["W"] for b in part],
                                                  [b["nW"] for b in part], [b["blocktype"] for b in part],
                                                  ampmax_in_first=part[0]["ampmax_in"], managed=bool(managed))
        assert not verdict.any()
        for k, b in enumerate(part):
            assert np.float32(ain[k]).tobytes() == np.float32(b["ampmax_in"]).tobytes(), (lo, k)
            assert np.float32(aout[k]).tobytes() == np.float32(b["ampmax_out"]).tobytes(), (lo, k)
            if managed:
                one, amp = an.encode_block(b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], ampmax_in=b["ampmax_in"], managed=True)
                assert pk[k] == one, (lo, k)
                assert b["packet"] in pk[k]       # the bitrate manager's choice is one of the fifteen
            else:
                assert pk[k] == [b["packet"]], (lo, k)
    # nothing to do is not an error
    pk, ain, aout, verdict = an.encode_blocks([], [], [], [], [])
    assert pk == [] and len(verdict) == 0
