"""The reference's own test for this path, cell by cell: test/test.c:30-75.

xiph/vorbis ships one test that exercises the encode path: for 1..8 channels x q = -0.05 .. 0.95 (eleven steps of .1,
accumulated in a float as `q+=.1` does) x six sample rates it encodes `gen_windowed_sine` (test/util.c:32-47: a
sine of period 32 under a Hann window over the first 1024 of 2048 samples, peak 0.95, the same samples in every
channel), decodes the result and requires the decoded peak within .15 - .1 q of 0.95 (`check_output`,
test/test.c:77-98).  oracle/ref_harness.c:ref_matrix_case restates write_vorbis_data_or_die /
read_vorbis_data_or_die (test/write_read.c) with the packets handed from encoder to decoder directly (libogg's page
framing is not in the reference tree).

  * CPU (`-m "not gpu"`): the unmodified reference passes its own grid through that restatement -- which pins the
    restatement -- all 528 cells.
  * GPU (`-m gpu`): the same grid through libvorbis_hybrid.so, whose mapping0_forward and block-switching detector
    are libvorbis_amd.so: every packet (headers included) byte-identical to the reference's, and the reference's
    own vorbis_synthesis + check_output on the GPU-made packets.
"""
import numpy as np
import pytest

from oracle import ref

RATES = [44100, 48000, 32000, 22050, 16000, 96000]   # test/test.c:37
DATA_LEN = 2048                                       # test/test.c:23


def gen_windowed_sine(length=DATA_LEN, maximum=0.95):
    """test/util.c:32-47, in its types: data[k] = (float)sin(..) then *= (double) window, rounded to float."""
    data = np.zeros(length, np.float32)
    half = length // 2
    k = np.arange(half, dtype=np.float64)
    s = np.sin(2.0 * k * np.pi * 1.0 / 32.0 + 0.4).astype(np.float32)
    w = np.float64(np.float32(maximum)) * (0.5 - 0.5 * np.cos(2.0 * np.pi * k / (half - 1)))
    data[:half] = (s.astype(np.float64) * w).astype(np.float32)
    return data


def q_steps():
    """float q=-.05; while(q<1.){ ...; q+=.1; }  (test/test.c:46-71): the float the reference actually passes"""
    out, q = [], np.float32(-.05)
    while np.float64(q) < 1.:
        out.append(float(q))
        q = np.float32(np.float64(q) + .1)
    return out


def check_output(data_in, q):
    """test/test.c:77-98 with allowable = .15f - .1f*q (:62)"""
    allowable = np.float32(.15) - np.float32(.1) * np.float32(q)
    max_abs = np.float32(np.abs(data_in).max())
    return (max_abs >= np.float32(np.float64(0.95) - np.float64(allowable)) and
            max_abs <= np.float32(np.float64(.95) + np.float64(allowable))), float(max_abs)


pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


def test_grid_shape():
    qs = q_steps()
    assert len(qs) == 11 and abs(qs[0] + .05) < 1e-6 and abs(qs[-1] - .95) < 1e-5
    d = gen_windowed_sine()
    assert d.shape == (DATA_LEN,) and not d[DATA_LEN // 2:].any() and 0.94 < np.abs(d).max() <= 0.95


@pytest.mark.parametrize("ch", range(1, 9))
def test_reference_passes_its_own_grid(ch):
    data = gen_windowed_sine()
    for q in q_steps():
        for rate in RATES:
            packets, dec, total = ref.matrix_case(ch, rate, q, data)
            assert total == DATA_LEN and len(packets) > 3
            ok, peak = check_output(dec, q)
            assert ok, "reference fails test/test.c cell ch=%d q=%.2f rate=%d: max_abs %f" % (ch, q, rate, peak)


@pytest.mark.gpu
@pytest.mark.skipif(not ref.hybrid_available(), reason="oracle/_ref/libvorbis_hybrid.so not built")
@pytest.mark.parametrize("ch", range(1, 9))
def test_gpu_backend_passes_the_reference_grid(ch):
    L = ref.lib(hybrid=True)
    assert hasattr(L, "vamd_encode_block") and hasattr(L, "_ve_envelope_search_cpu")   # the GPU bindings are linked in
    data = gen_windowed_sine()
    cells = 0
    for q in q_steps():
        for rate in RATES:
            want, wdec, _ = ref.matrix_case(ch, rate, q, data)
            got, gdec, total = ref.matrix_case(ch, rate, q, data, hybrid=True)
            tag = "ch=%d q=%.2f rate=%d" % (ch, q, rate)
            assert len(got) == len(want), tag
            for k, (a, b) in enumerate(zip(want, got)):
                assert a == b, "%s: packet %d differs (%d vs %d bytes)" % (tag, k, len(a), len(b))
            assert total == DATA_LEN and np.array_equal(wdec, gdec), tag
            ok, peak = check_output(gdec, q)
            assert ok, "%s: decoded max_abs %f outside .95 +- allowable" % (tag, peak)
            cells += 1
    assert cells == 66


@pytest.mark.gpu
def test_every_grid_cell_takes_the_packet_path():
    """Which setups does the binding serve with GPU-assembled packets (vamd_packet_capacity() > 0: vamd_encode_block,
    integration/mapping0_vamd.c) and which through vamd_write_packet (the reference's own floor1_encode / res*_forward on
    the host)?  Every cell of test/test.c's grid -- 1..8 channels x 11 qualities x 6 rates -- is asked; the answer is
    written down here and in INTEGRATION.md: all of them take the packet path, both block sizes."""
    import vorbis_amd
    host_cells = []
    cells = 0
    for ch in range(1, 9):
        for q in q_steps():
            for rate in RATES:
                e = ref.RefEncoder(ch, rate, q)
                an = vorbis_amd.Analyzer(e.pack_setup(), 0)
                caps = [an.packet_capacity(W) for W in (0, 1)] + [an.residue_capacity(W) for W in (0, 1)]
                an.close()
                e.close()
                cells += 1
                if min(caps) <= 0:
                    host_cells.append((ch, rate, round(q, 2), caps))
    assert cells == 528
    assert not host_cells, "cells whose packets the host writes (vamd_write_packet): %s" % host_cells
