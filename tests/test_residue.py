"""The residue back-end's numeric half (res*_class + the lattice-VQ search of res*_forward, reference
lib/res0.c:322-640,729-809).

What pins what:
  * the reference's own decisions are tapped without editing it: lib/res0.c is compiled with
    -Dvorbis_book_encode=ref_tap_book_encode (oracle/Makefile), so every codebook entry it emits passes
    through the harness on its way to the real bit-writer; classes come from the extern class function.
    tests/golden/blocks_*.npz carry both for every fixture block (tools/make_golden.py).
  * oracle/port (sequential restatement) and the kernel bodies compiled for the host must reproduce
    them exactly -- CPU suite; the HIP library must too, per block and batched -- GPU suite.
  * end to end, tests/test_packets.py / tests/test_gpu_dropin.py: the packets assembled from these
    entries are byte-identical to the reference's.
"""
import ctypes as C
import os

import numpy as np
import pytest

import vorbis_amd
from oracle import port, ref
from tests import checker, golden_io

ROOT = checker.ROOT
NAMES = list(checker.SETUPS)


def blob_of(name):
    return np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % name), dtype=np.uint8)


def same(a, b):
    return np.array_equal(a["res_class"], b["res_class"]) and np.array_equal(a["res_entries"], b["res_entries"])


# ------------------------------------------------------------------------------------------
# CPU suite
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", NAMES)
def test_port_and_kernel_bodies_match_golden(name):
    from tests.emul.emul import Emul
    blocks, _, _ = golden_io.load(name)
    p, em = port.PortEncoder(blob_of(name)), Emul(blob_of(name))
    coded = 0
    for b in blocks:
        args = (b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])
        assert same(b, p.tap_block(*args)), ("port", b["W"], b["blocktype"])
        assert same(b, em.analyze_block(*args)), ("kernel bodies", b["W"], b["blocktype"])
        coded += len(b["res_entries"])
    assert coded > 1000                                   # the fixtures do exercise the search
    assert any(len(b["res_class"]) == 0 for b in blocks)  # ... and the nothing-to-code case (silence)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("quality", [-0.1, 0.1, 0.4, 0.7, 1.0])
def test_against_reference_random_blocks(quality):
    """Every libvorbisenc quality (different book sets, class tables, residue ends), both block sizes;
    loud blocks push values off the lattice books' populated entries (the exhaustive-search branch)."""
    from tests.emul.emul import Emul
    r = ref.RefEncoder(2, 44100, quality)
    blob = r.pack_setup()
    p = port.PortEncoder(blob)
    em = Emul(blob)
    rng = np.random.default_rng(int(quality * 100) + 77)
    for it in range(6):
        W = 0 if it == 5 else 1
        n = r.blocksize(W)
        amp = [0.5, 0.01, 1.0, 1e-4, 0.9, 0.7][it]
        pcm = ((rng.random((2, n), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        if it == 4:
            pcm[:, ::7] *= -1.0  # spiky: large isolated residues
        args = (pcm, W, W, W, 1 if W else 0)
        a = r.tap_block(*args)
        assert a["packet_matches_real"]
        assert same(a, p.tap_block(*args)), ("port", it)
        g = em.analyze_block(*args)
        assert "res_class" not in g or same(a, g), ("kernel bodies", it)  # (128-partition residues stay on the host)


def test_residue_tables_in_blob_are_consistent():
    """Capacity bound: no class can emit more vectors per partition than the row length assumes."""
    from tests.emul.emul import Emul
    for name in NAMES:
        em = Emul(blob_of(name))
        for W in (0, 1):
            cap = em.L.emul_residue_capacity(em.h, W)
            assert cap > 0, (name, W)        # all four shipped setups are covered
            assert cap <= 512 * 64


def test_packed_residue_table_matches_the_blob():
    """The (class, stage) rows k_residue_chunks / k_pack_waves keep in LDS (ResP::fast, built at bind time) against the blob's
    residue and codebook tables field by field, and ResP::chunked against its conditions -- every shipped setup, every
    libvorbisenc rate family through the reference where it is at hand.  The stereo setups must take the chunked search."""
    from tests.emul.emul import Emul
    import ctypes as C
    blobs = [(name, blob_of(name)) for name in NAMES]
    if ref.available():
        for ch, rate, q in ((2, 8000, 0.3), (2, 22050, 0.5), (2, 32000, 0.1), (2, 48000, 0.9), (2, 96000, 0.5), (1, 44100, 0.4), (6, 44100, 0.3)):
            blobs.append(("%dch %d q%.1f" % (ch, rate, q), ref.RefEncoder(ch, rate, q).pack_setup()))
    for name, blob in blobs:
        em = Emul(blob)
        em.L.emul_residue_fast_check.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        for W in (0, 1):
            for sm in range(em.L.emul_submaps(em.h, W)):
                chunked = C.c_int(-1)
                assert em.L.emul_residue_fast_check(em.h, W, sm, C.byref(chunked)) == 0, (name, W, sm)
                if name.startswith("44k_stereo") or name.startswith("2ch"):
                    assert chunked.value == 1, (name, W, sm)


# ------------------------------------------------------------------------------------------
# GPU suite (through the C ABI)
# ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_block_api_matches_golden(name):
    blocks, _, _ = golden_io.load(name)
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    for b in blocks:
        g = an.analyze_block(b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])
        assert same(b, g), (b["W"], b["blocktype"])
        assert np.array_equal(g["iwork"], b["iwork"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["44k_stereo_q4", "44k_stereo_q9", "44k_stereo_q1", "44k_mono_q5"])
def test_gpu_batch_matches_checker(name):
    import torch
    chk = checker.Checker(name)
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    ch = an.channels
    rng = np.random.default_rng(31)
    for W, nb in ((1, 48), (0, 32)):
        n = an.blocksizes[W]
        amp = np.array([0.5, 0.01, 1.0, 0.0, 0.9])[np.arange(nb) % 5, None, None]
        pcm = ((rng.random((nb, ch, n), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        o = an.analyze(torch.from_numpy(pcm).cuda(), W=W, lW=W, nW=W, blocktype=1 if W else 0,
                       want=("iwork", "nonzero", "res_class", "res_entries", "res_count"))
        torch.cuda.synchronize()
        cnt = o["res_count"].cpu().numpy()
        cls = o["res_class"].cpu().numpy()
        ent = o["res_entries"].cpu().numpy().view(np.uint16)
        assert cnt[:, 1].max() <= an.residue_capacity(W)
        for k in range(nb):
            a = chk.tap_block(pcm[k], W, W, W, 1 if W else 0)
            assert np.array_equal(a["res_class"], cls[k, :cnt[k, 0]]), (W, k)
            assert np.array_equal(a["res_entries"], ent[k, :cnt[k, 1]]), (W, k)
        assert (cnt[:, 0] == 0).any() and (cnt[:, 1] > 0).any()


@pytest.mark.gpu
def test_gpu_residue_argument_errors():
    import torch
    an = vorbis_amd.Analyzer(blob_of("44k_stereo_q4"), 0)
    pcm = torch.zeros((2, 2, 2048), device="cuda")
    outs = an.alloc_outputs(1, 2, ("res_class", "res_entries", "res_count"))
    with pytest.raises(vorbis_amd.VamdError) as e:   # residue outputs need the full analysis
        an.analyze(pcm, level=vorbis_amd.LEVEL_PSY, outs=outs)
    assert e.value.code == -131
    del outs["res_count"]                            # the three outputs go together
    with pytest.raises(vorbis_amd.VamdError) as e:
        an.analyze(pcm, outs=outs)
    assert e.value.code == -131
    assert an.L.vamd_residue_capacity(None, 1) == 0
