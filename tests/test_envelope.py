"""The block-switching detector (_ve_envelope_search / _ve_amp, reference lib/envelope.c:89-262;
SURVEY.md 8f rank 1).

What pins what:
  * tests/golden/envelope_*.npz were written by the reference's own _ve_envelope_search
    (tools/make_golden_envelope.py): PCM exactly as the detector saw it, the marks it set, its final
    filter state.
  * the plain-C restatement oracle/port (running-state form) and the product's kernel bodies compiled
    for the host (tests/emul; replay form) must both reproduce those marks and that state bit for bit
    -- CPU suite;
  * the HIP library must too, in one call, in ragged chunks with the state carried across, and for a
    batch of streams -- GPU suite, through the C ABI.
"""
import ctypes as C
import os

import numpy as np
import pytest

import vorbis_amd
from vorbis_amd import EnvelopeState, envelope_marks
from oracle import port, ref
from tests import checker

ROOT = checker.ROOT
GOLDEN = ("44k_stereo_q4", "44k_mono_q5")


def blob_of(name):
    return np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % name), dtype=np.uint8)


def golden(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "envelope_%s.npz" % name))
    return dict(pcm=z["pcm"], marks=z["marks"], steps=int(z["steps"][0]), stretch=int(z["stretch"][0]),
                near=z["near"], amp=z["amp"])


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def gated(ch, frames, seed, period=5000, on=700):
    rng = np.random.default_rng(seed)
    t = np.arange(frames)
    g = np.where((t % period) < on, 0.6, 0.004).astype(np.float32)
    return ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * g).astype(np.float32)


def check_state(st, want, ch):
    """EnvelopeState (history form) against the reference's rings (already rolled oldest-first)."""
    assert st.stretch == want["stretch"]
    amp = np.array(st.amp_hist, np.float32).reshape(8, 16, 8)[:ch]
    near = np.array(st.near_hist, np.float32).reshape(8, 30)[:ch]
    # reference keeps 17 amplitudes / 15 near-DC terms; the histories keep 16 / 30
    assert np.array_equal(bits(amp[:, :, :7].transpose(0, 2, 1)), bits(want["amp"][:, :, 1:]))
    assert np.array_equal(bits(near[:, 15:]), bits(want["near"]))


# ------------------------------------------------------------------------------------------
# CPU suite
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", GOLDEN)
def test_port_matches_golden(name):
    g = golden(name)
    flags, st = port.PortEncoder(blob_of(name)).envelope_steps(g["pcm"], g["steps"])
    assert np.array_equal(envelope_marks(flags)[:g["steps"] + 2], g["marks"])
    assert st.stretch == g["stretch"]
    assert flags.max() <= 7 and not np.any((flags & 1) != ((flags >> 2) & 1))  # 1 and 4 always travel together


@pytest.mark.parametrize("name", GOLDEN)
def test_kernel_bodies_match_golden(name):
    from tests.emul.emul import Emul
    g = golden(name)
    ch = g["pcm"].shape[0]
    em = Emul(blob_of(name))
    st = EnvelopeState()
    flags = em.envelope_search(g["pcm"], g["steps"], st)
    assert np.array_equal(envelope_marks(flags)[:g["steps"] + 2], g["marks"])
    check_state(st, g, ch)
    assert st.steps == g["steps"]
    # ragged chunks with the state carried across calls give the same flags
    st2 = EnvelopeState()
    out, j = [], 0
    for k in (1, 2, 13, 15, 16, 31, 64, 10 ** 6):
        k = min(k, g["steps"] - j)
        if k <= 0:
            break
        out.append(em.envelope_search(g["pcm"][:, j * 64:], k, st2))
        j += k
    assert j == g["steps"] and np.array_equal(np.concatenate(out), flags)
    check_state(st2, g, ch)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,seed", [("44k_stereo_q4", 1), ("44k_stereo_q9", 2), ("44k_stereo_q1", 3), ("44k_mono_q5", 4)])
def test_port_and_bodies_match_reference_streams(name, seed):
    """Longer streams, every setup, fed to the reference in two instalments (its state persists)."""
    from tests.emul.emul import Emul
    ch, rate, q = checker.SETUPS[name]
    x = gated(ch, 50000, seed)
    e = ref.RefEncoder(ch, rate, q)
    e.envelope_feed(x[:, :17000])
    o = e.envelope_feed(x[:, 17000:])
    assert o["marks"].sum() > 10
    flags, pst = port.PortEncoder(blob_of(name)).envelope_steps(o["pcm"], o["steps"])
    assert np.array_equal(envelope_marks(flags)[:o["steps"] + 2], o["marks"])
    st = EnvelopeState()
    flags2 = Emul(blob_of(name)).envelope_search(o["pcm"], o["steps"], st)
    assert np.array_equal(flags2, flags)
    check_state(st, o, ch)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,seed", [("44k_stereo_q4", 5), ("44k_stereo_q9", 6), ("44k_mono_q5", 7)])
def test_blockout_walk_matches_reference(name, seed):
    """k_blockout.h (the walk k_plan_streams runs per stream) compiled for the host: from the detector's flags over the
    encoder's PCM buffer to the block list -- sizes, window flags, block types, positions -- of the reference's own
    vorbis_analysis_blockout() on the same stream (lib/block.c:534-693)."""
    from tests.emul.emul import Emul
    ch, rate, q = checker.SETUPS[name]
    frames = 60000
    x = gated(ch, frames, seed)
    e = ref.RefEncoder(ch, rate, q)
    for k in range(0, frames, 1024):   # the application's feed size: the start-of-stream extrapolation sees this much
        o = e.envelope_feed(x[:, k:k + 1024])
    blocks = ref.RefEncoder(ch, rate, q).encode_stream(x)
    em = Emul(blob_of(name))
    nsamples = o["pcm"].shape[1]
    nsteps = nsamples // 64 - 4
    flags = em.envelope_search(o["pcm"], nsteps, EnvelopeState())
    kind, begin = em.plan_stream(flags, nsamples)
    assert 10 < len(kind) <= len(blocks) and len(blocks) - len(kind) <= 12
    assert len(set(int(k) & 1 for k in kind)) == 2, "the stream should switch block sizes"
    for k in range(len(kind)):
        b = blocks[k]
        assert (int(kind[k]) & 1, (int(kind[k]) >> 1) & 1, (int(kind[k]) >> 2) & 1, (int(kind[k]) >> 3) & 1) == \
            (b["W"], b["lW"], b["nW"], b["blocktype"]), k
        n = b["pcm"].shape[1]
        assert np.array_equal(o["pcm"][:, begin[k]:begin[k] + n], b["pcm"]), k


def test_envelope_state_layout_matches_header():
    """ctypes mirror == the C struct of include/vorbis_amd.h (sizes the device-side state arrays)."""
    hdr = open(os.path.join(ROOT, "include", "vorbis_amd.h")).read()
    assert "#define VAMD_VE_NEAR_HIST 30" in hdr and "#define VAMD_VE_AMP_HIST  16" in hdr
    assert "#define VAMD_MAX_CH        8" in open(os.path.join(ROOT, "include", "vamd_setup.h")).read()
    assert C.sizeof(EnvelopeState) == 8 + 4 + 4 + 8 * 30 * 4 + 8 * 16 * 8 * 4
    assert EnvelopeState.steps.offset == 0 and EnvelopeState.stretch.offset == 8
    assert EnvelopeState.near_hist.offset == 16 and EnvelopeState.amp_hist.offset == 16 + 960


# ------------------------------------------------------------------------------------------
# GPU suite (through the C ABI)
# ------------------------------------------------------------------------------------------
def _checker_stream(name, frames, seed):
    """(pcm as the detector sees it, steps, wanted flags or None, wanted marks, reference state or None)."""
    ch, rate, q = checker.SETUPS[name]
    x = gated(ch, frames, seed)
    if ref.available():
        o = ref.RefEncoder(ch, rate, q).envelope_feed(x)
        return o["pcm"], o["steps"], o["marks"], o
    pcm = np.concatenate([np.zeros((ch, 1024), np.float32), x], axis=1)  # centre padding, no pre-extrapolation
    steps = pcm.shape[1] // 64 - 4
    flags, _ = port.PortEncoder(blob_of(name)).envelope_steps(pcm, steps)
    return pcm, steps, envelope_marks(flags)[:steps + 2], None


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN)
def test_gpu_matches_golden(name):
    g = golden(name)
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    flags, st = an.envelope_search(g["pcm"], g["steps"])
    assert np.array_equal(envelope_marks(flags)[:g["steps"] + 2], g["marks"])
    check_state(st, g, g["pcm"].shape[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name,seed", [("44k_stereo_q4", 11), ("44k_stereo_q9", 12), ("44k_stereo_q1", 13), ("44k_mono_q5", 14)])
def test_gpu_stream_whole_and_chunked(name, seed):
    pcm, steps, marks, o = _checker_stream(name, 120000, seed)
    assert marks.sum() > 20
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    flags, st = an.envelope_search(pcm, steps)
    assert np.array_equal(envelope_marks(flags)[:steps + 2], marks)
    if o is not None:
        check_state(st, o, pcm.shape[0])
    # the way the binding calls it: a block's worth of steps at a time, state carried
    st2 = EnvelopeState()
    out, j, k = [], 0, 0
    sizes = (1, 3, 16, 17, 32, 5, 64, 250)
    while j < steps:
        n = min(sizes[k % len(sizes)], steps - j)
        f, st2 = an.envelope_search(pcm[:, j * 64:j * 64 + (n - 1) * 64 + 128], n, st2)
        out.append(f)
        j += n
        k += 1
    assert np.array_equal(np.concatenate(out), flags)
    assert bytes(st2) == bytes(st)


@pytest.mark.gpu
def test_gpu_batch_of_streams():
    import torch
    name = "44k_stereo_q4"
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    streams = [_checker_stream(name, 40000, 100 + s) for s in range(5)]
    steps = min(s[1] for s in streams)
    ln = (steps - 1) * 64 + 128
    pcm = torch.from_numpy(np.stack([s[0][:, :ln] for s in streams])).cuda()
    ret, states = an.envelope_search_batch(pcm, steps)
    ret = ret.cpu().numpy()
    for k, s in enumerate(streams):
        assert np.array_equal(envelope_marks(ret[k])[:steps], s[2][:steps]), k
    # second half of a longer run through the batch entry point, states carried on the device
    half = steps // 2
    ln1 = (half - 1) * 64 + 128
    r1, st = an.envelope_search_batch(pcm[:, :, :ln1].contiguous(), half)
    r2, st = an.envelope_search_batch(pcm[:, :, half * 64:].contiguous(), steps - half, states=st)
    assert np.array_equal(np.concatenate([r1.cpu().numpy(), r2.cpu().numpy()], axis=1), ret)
    assert torch.equal(st, states)


@pytest.mark.gpu
@pytest.mark.parametrize("name,ragged", [("44k_stereo_q4", 0), ("44k_mono_q5", 7), ("44k_stereo_q9", 37)])
def test_gpu_big_batch_takes_the_tiled_kernels(name, ragged):
    """Above 65 536 step-streams the band amplitudes and the trigger bits come from LDS-tiled kernels (k_env_amp_tiled: 32
    steps of a channel per workgroup; k_env_bits_tiled: 64 steps of a stream per wave); below it, from the thread-per-item
    forms.  The same streams, gated so that both flags fire, once as one big batch and once stream by stream, with step
    counts that leave ragged last tiles, and in two calls with the state carried: identical flags and identical states."""
    import torch
    an = vorbis_amd.Analyzer(blob_of(name), 0)
    ch = an.channels
    ns, steps = 96, 700 + ragged
    ln = (steps - 1) * 64 + 128
    rng = np.random.default_rng(5150 + ragged)
    t = np.arange(ln)
    gate = np.where(((t[None, :] + 997 * np.arange(ns)[:, None]) % (5000 + 13 * np.arange(ns)[:, None])) < 500, 0.6, 0.001)
    pcm = ((rng.random((ns, ch, ln), dtype=np.float32) - 0.5) * gate[:, None, :]).astype(np.float32)
    dev = torch.from_numpy(pcm).cuda()
    assert ns * steps > 65536
    big, st_big = an.envelope_search_batch(dev, steps)
    big = big.cpu().numpy()
    assert (big & 1).any() and (big & 2).any()
    for k in range(0, ns, 7):
        one, st_one = an.envelope_search_batch(dev[k:k + 1].contiguous(), steps)      # 700 step-streams: the small kernels
        assert np.array_equal(one.cpu().numpy()[0], big[k]), k
        assert torch.equal(st_one[0], st_big[k]), k
    first = 333
    r1, st = an.envelope_search_batch(dev[:, :, :(first - 1) * 64 + 128].contiguous(), first)
    r2, st = an.envelope_search_batch(dev[:, :, first * 64:].contiguous(), steps - first, states=st)
    assert np.array_equal(np.concatenate([r1.cpu().numpy(), r2.cpu().numpy()], axis=1), big)
    assert torch.equal(st, st_big)


@pytest.mark.gpu
def test_gpu_envelope_argument_errors():
    an = vorbis_amd.Analyzer(blob_of("44k_stereo_q4"), 0)
    assert an.envelope_geometry() == (128, 64)
    st = EnvelopeState()
    ret = np.zeros(4, np.uint8)
    assert an.L.vamd_envelope_search(an.h, None, 4, C.byref(st), ret.ctypes.data) == -131   # OV_EINVAL
    assert an.L.vamd_envelope_search(an.h, None, 0, C.byref(st), ret.ctypes.data) == 0      # nothing to do
    assert an.L.vamd_envelope_search_batch(an.h, None, 0, 0, -1, 4, None, None) == -131
