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
    for it in range(10):
        amp = [0.5, 0.003, 1.0, 0.1, 6.0][it % 5]
        pcm = ((rng.random((ch, 2048), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
        if it % 5 == 4:   # a tone above full scale: the block's spectral maximum is over 0 dB and is clamped (lib/mapping0.c:345)
            pcm += (amp * np.sin(np.arange(2048) * 0.05)).astype(np.float32)[None, :]
        a = chk.tap_block(pcm, ampmax_in=-9999.0 if it % 2 else -40.0)
        g = em.analyze_block(pcm, ampmax_in=-9999.0 if it % 2 else -40.0)
        assert checker.compare_block(a, g, em.L and 29, verbose=True) == 0


@pytest.mark.parametrize("name", ["44k_stereo_q4", "44k_51_q3", "44k_mono_q5", "44k_stereo_q9"])
def test_fit_work_list_covers_the_reference_ranges(name):
    """k_floor's accumulate_fit reads a static list of (chunk, interval) records (vamd_derive.h derive_fit_segments):
    every bin must be summed into exactly the intervals lib/floor1.c:601-608 sums it into, the shared posts twice."""
    em = Emul(np.fromfile(os.path.join(checker.ROOT, "vorbis_amd", "data", "setup_%s.bin" % name), dtype=np.uint8))
    assert em.fit_segments_mismatches() == 0


def test_div_magic_is_exact_on_the_floor_lines_domain():
    """k_floor's line walks divide by x1 - x0 with one multiply (vamd_wave.h div_magic, table derive_div_magic)."""
    em = Emul(np.fromfile(os.path.join(checker.ROOT, "vorbis_amd", "data", "setup_44k_stereo_q4.bin"), dtype=np.uint8))
    assert em.div_magic_mismatches() == 0


def test_quant_energy_equals_the_fp64_expression():
    """k_couple's +-rint(sqrt(ve)) without fp64 (k_couple.h quant_energy) against lib/psy.c:958-962 as written."""
    em = Emul(np.fromfile(os.path.join(checker.ROOT, "vorbis_amd", "data", "setup_44k_stereo_q4.bin"), dtype=np.uint8))
    assert em.quant_energy_mismatches() == 0


def test_couple_estimates_decide_like_the_divisions():
    """k_couple's estimate-then-verify forms (k_couple.h chan_bin_sure / couple_bin_sure): wherever they call a bin sure
    it equals the exact form, with the hardware's estimates anywhere inside their one-ulp promise; and the exact path
    is the rare one."""
    em = Emul(np.fromfile(os.path.join(checker.ROOT, "vorbis_amd", "data", "setup_44k_stereo_q4.bin"), dtype=np.uint8))
    bad, ppm = em.couple_estimate_mismatches()
    assert bad == 0
    assert ppm < 400, ppm  # (|m| / f log-uniform up to 1024: the share grows with the magnitude)


def test_chunked_chase_equals_serial_walk():
    """k_tone_chase_wave's algorithm on the host: the stack walk of seed_chase (lib/psy.c:454-487) cut into 64 chunks
    with cold starts, entry/exit state verification and repair rounds gives the serial walk's survivor list on every
    kind of seed vector; ordinary vectors verify at once or after a round or two, stretches of equal values take a few more, and a block with a
    very long one is handed to the serial walk (which the kernel then runs) -- never a different list."""
    from tests.emul.emul import Emul
    em = Emul(np.fromfile(os.path.join(checker.ROOT, "vorbis_amd", "data", "setup_44k_stereo_q4.bin"), dtype=np.uint8))
    rng = np.random.default_rng(2024)
    for trial in range(500):
        n = int(rng.choice([784, 784, 784, 400, 97, 1024, 33, 64, 65]))
        kind = trial % 10
        x = (rng.random(n) * 60 - 30).astype(np.float32)                  # white
        if kind == 1:
            x = np.round(x / 6) * 6                                         # heavy ties
        elif kind == 2:
            x = np.cumsum(rng.standard_normal(n)).astype(np.float32)        # slow drifts
        elif kind == 3:
            x[rng.random(n) < 0.7] = -9999.0                                # mostly empty lines
        elif kind == 4:
            x = np.sort(x)                                                  # rising
        elif kind == 5:
            x = np.sort(x)[::-1].copy()                                     # falling
        elif kind == 6:
            x[:] = 3.0                                                      # flat throughout
        elif kind == 7:
            x = (np.sin(np.arange(n) * rng.random() * 3) * 20).astype(np.float32)
        elif kind == 8:
            x[n // 2:] = -9999.0                                            # nothing reached the upper half
        elif kind == 9:
            x[3 * n // 4:] = -9999.0
        for L in (8, 8, 4, 16):
            same, accepted, ns, rounds = em.chase_compare(x.astype(np.float32), L)
            assert same, (trial, kind, n, L)
            assert ns > 0
            if kind in (0, 1, 2, 4, 5, 7):   # ordinary vectors: accepted, after a repair round or a few at most (ties, wide windows)
                assert accepted and rounds <= 6, (trial, kind, n, L, rounds)
            if kind == 9:
                assert accepted, (trial, n, L, rounds)
            if kind == 6 and n >= 784:
                assert not accepted and rounds < 0     # routed to the serial walk before any chunk is walked
    # the register form (eight lines per window: k_tone_seed_chase) at the other line counts it meets -- 585 (short blocks),
    # and up to 2048, where a chunk is 32 lines, the most its popped-mask holds
    rng = np.random.default_rng(4)
    for trial in range(120):
        n = int(rng.choice([585, 777, 1999, 2048, 2047, 1025]))
        x = (rng.random(n) * 60 - 30).astype(np.float32)
        if trial % 4 == 1:
            x = np.round(x / 6) * 6
        elif trial % 4 == 2:
            x[rng.random(n) < 0.6] = -9999.0
        elif trial % 4 == 3:
            x = (np.sin(np.arange(n) * rng.random() * 3) * 20).astype(np.float32)
        same, accepted, ns, rounds = em.chase_compare(x.astype(np.float32), 8)
        assert same and ns > 0, (trial, n)
        if trial % 4 != 2:
            assert accepted, (trial, n, rounds)
