"""CPU suite, part 2: the product's kernel bodies (vorbis_amd/csrc/k_*.h) compiled by the host
compiler as a test build (tests/emul) must agree bit-for-bit with the oracle.  This checks the
arithmetic and the parallel re-formulation (flattened butterflies, closed-form Bresenham,
precomputed run/span tables, order-free seed scatter) without a GPU; the GPU suite then only has
to confirm that the 64-lane execution matches."""
import os

import numpy as np
import pytest

from tests import checker, golden_io
from tests.emul.emul import Emul

ROOT = checker.ROOT


def blob_of(name):
    return np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % name), dtype=np.uint8)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("name", list(checker.SETUPS))
def test_bodies_match_golden(name):
    blocks, posts, fn = golden_io.load(name)
    em = Emul(blob_of(name))
    for W in (0, 1):
        assert np.array_equal(bits(em.mdct_forward(W, fn["mdct%d_in" % W])), bits(fn["mdct%d_out" % W]))
    for b in blocks:
        g = em.analyze_block(b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])
        ref = dict(b)
        assert checker.compare_block(ref, g, posts[b["W"]], verbose=True) == 0


@pytest.mark.parametrize("name", ["44k_stereo_q4", "44k_stereo_q1", "44k_mono_q5"])
def test_bodies_match_oracle_random(name):
    chk = checker.Checker(name)
    em = Emul(blob_of(name))
    ch = checker.SETUPS[name][0]
    rng = np.random.default_rng(77)
    for it in range(8):
        amp = [0.5, 0.003, 1.0, 0.1][it % 4]
        pcm = ((rng.random((ch, 2048), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        a = chk.tap_block(pcm, ampmax_in=-9999.0 if it % 2 else -40.0)
        g = em.analyze_block(pcm, ampmax_in=-9999.0 if it % 2 else -40.0)
        assert checker.compare_block(a, g, em.L and 29, verbose=True) == 0
