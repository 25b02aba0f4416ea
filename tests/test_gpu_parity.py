"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle.

Everything is bit-exact: the integer outputs by definition, the float outputs because the
kernels evaluate the reference's own expression trees in fp32/fp64 without contraction (the
1e-5 relative tolerance of BASELINE.json is therefore met with margin zero).
"""
import os

import numpy as np
import pytest

from tests import checker, golden_io

pytestmark = pytest.mark.gpu
ROOT = checker.ROOT
ALL = ("mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct", "posts", "post_valid", "ilogmask",
       "iwork", "nonzero", "local_ampmax", "ampmax_out")


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def analyzer(name):
    import vorbis_amd
    return vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(name), device=0)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def run_blocks(torch, an, blocks, W):
    sel = [b for b in blocks if b["W"] == W]
    if not sel:
        return [], {}
    P = torch.from_numpy(np.stack([b["pcm"] for b in sel])).cuda()
    dv = lambda k, dt: torch.tensor([b[k] for b in sel], dtype=dt).cuda()  # noqa: E731
    outs = an.analyze(P, W=W, lW=dv("lW", torch.int32), nW=dv("nW", torch.int32),
                      blocktype=dv("blocktype", torch.int32), ampmax_in=dv("ampmax_in", torch.float32), want=ALL)
    torch.cuda.synchronize()
    return sel, {k: v.cpu().numpy() for k, v in outs.items()}


def test_library_is_the_hip_build(torch_mod):
    import vorbis_amd
    L = vorbis_amd.load_library()
    assert os.path.samefile(vorbis_amd.library_path(), os.path.join(ROOT, "vorbis_amd", "libvorbis_amd.so"))
    assert L.vamd_create_abi and L.vamd_analyze_batch


@pytest.mark.parametrize("name", list(checker.SETUPS))
def test_golden_blocks(torch_mod, name):
    """Committed fixtures from the reference: long/short/transition windows, impulse/padding block
    types, chained ampmax, silence, a pure tone; q 0.1 exercises noise normalisation's sort."""
    blocks, posts, fn = golden_io.load(name)
    an = analyzer(name)
    for W in (0, 1):
        x = torch_mod.from_numpy(fn["mdct%d_in" % W][None, :].copy()).cuda()
        y = an.mdct_forward(W, x).cpu().numpy()[0]
        assert np.array_equal(bits(y), bits(fn["mdct%d_out" % W]))
        sel, outs = run_blocks(torch_mod, an, blocks, W)
        for i, b in enumerate(sel):
            got = {k: v[i] for k, v in outs.items()}
            assert checker.compare_block(b, got, posts[W], verbose=True) == 0


@pytest.mark.parametrize("name", ["44k_stereo_q4", "44k_stereo_q9", "44k_stereo_q1", "44k_mono_q5"])
def test_random_blocks_vs_oracle(torch_mod, name):
    torch = torch_mod
    chk = checker.Checker(name)
    an = analyzer(name)
    ch = an.channels
    rng = np.random.default_rng(31337)
    nb = 96
    amps = np.array([0.5, 0.01, 1.0, 1e-4, 0.2, 0.05])[np.arange(nb) % 6].astype(np.float32)
    pcm = ((rng.random((nb, ch, 2048), dtype=np.float32) - 0.5) * 2 * amps[:, None, None]).astype(np.float32)
    if ch == 2:
        pcm[5::7, 1] = pcm[5::7, 0] * 0.5     # correlated channels
        pcm[3::11, 1] = 0                       # one silent channel: zero floor + coupling fix-up
    pcm[17] = 0                                 # digital silence
    pcm[23::24] *= 12.0                         # far over full scale: local ampmax above 0 dB, clamped (lib/mapping0.c:345)
    pcm[23::24] += (6.0 * np.sin(np.arange(2048) * 0.07)).astype(np.float32)[None, None, :]
    amp_in = np.where(np.arange(nb) % 3 == 0, -9999.0, -35.0).astype(np.float32)
    outs = an.analyze(torch.from_numpy(pcm).cuda(), W=1, ampmax_in=torch.from_numpy(amp_in).cuda(), want=ALL)
    torch.cuda.synchronize()
    outs = {k: v.cpu().numpy() for k, v in outs.items()}
    bad = 0
    for b in range(nb):
        ref = chk.tap_block(pcm[b], ampmax_in=float(amp_in[b]))
        bad += checker.compare_block(ref, {k: v[b] for k, v in outs.items()}, an.posts[1], verbose=bad < 5)
    assert bad == 0, "checker=%s" % chk.kind


@pytest.mark.parametrize("nb", [1, 2, 31, 32, 33, 63, 65, 129])
def test_batch_sizes_around_the_launch_thresholds(torch_mod, nb):
    """The launch sequence changes shape with the batch: up to 32 blocks (64 channel-blocks) the tone chain stays on the
    caller's stream and the chase runs a wave per block, above that it forks onto the side stream; persistent kernels get
    fewer workgroups than CUs; the block ampmax is formed inside the tone seed (no launch of its own).  Every size lands
    on the same bytes, ampmax_out included, for both ways an incoming ampmax can arrive (array / uniform)."""
    torch = torch_mod
    name = "44k_stereo_q4"
    chk = checker.Checker(name)
    an = analyzer(name)
    rng = np.random.default_rng(4000 + nb)
    pcm = ((rng.random((nb, 2, 2048), dtype=np.float32) - 0.5) * 2 * (10.0 ** rng.uniform(-3, 0, (nb, 1, 1)))).astype(np.float32)
    amp_in = np.where(np.arange(nb) % 2 == 0, -9999.0, -20.0).astype(np.float32)
    outs = an.analyze(torch.from_numpy(pcm).cuda(), W=1, ampmax_in=torch.from_numpy(amp_in).cuda(), want=ALL)
    uni = an.analyze(torch.from_numpy(pcm).cuda(), W=1, ampmax_in=-20.0, want=("ampmax_out", "iwork"))
    torch.cuda.synchronize()
    outs = {k: v.cpu().numpy() for k, v in outs.items()}
    bad = 0
    for b in range(nb):
        ref = chk.tap_block(pcm[b], ampmax_in=float(amp_in[b]))
        bad += checker.compare_block(ref, {k: v[b] for k, v in outs.items()}, an.posts[1], verbose=bad < 3)
        if amp_in[b] == -20.0:
            assert np.float32(uni["ampmax_out"][b].item()) == np.float32(ref["ampmax_out"])
            assert np.array_equal(uni["iwork"][b].cpu().numpy(), ref["iwork"])
    assert bad == 0, "checker=%s" % chk.kind


@pytest.mark.parametrize("name", ["44k_stereo_q4", "44k_stereo_q9", "44k_mono_q5"])
def test_couple_estimate_band_does_not_show(torch_mod, name, monkeypatch):
    """k_couple decides a bin from an estimate of |m| / floor wherever that provably equals the reference's divisions
    and sends the rest through them (k_couple.h chan_bin_sure).  VAMD_COUPLE_BAND_LOG2 (read at vamd_create) widens
    the margin: at 1 every quad takes the exact path, at -6 about every other wave does, the default (-21) almost
    none -- and the residue must be the same in all three (the default is also what every other parity test runs)."""
    torch = torch_mod
    rng = np.random.default_rng(99)
    nb = 256
    ch = 1 if "mono" in name else 2
    amp = (10.0 ** rng.uniform(-3, 0.5, (nb, 1, 1))).astype(np.float32)
    pcm = ((rng.random((nb, ch, 2048), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
    t = np.arange(2048, dtype=np.float32)
    for k in range(0, nb, 2):  # every other block: two tones far above the noise (large quantised values)
        pcm[k] += (0.4 * np.sin(t * (0.02 + 0.003 * k)) + 0.2 * np.sin(t * (0.31 + 0.001 * k))).astype(np.float32)[None, :]
    P = torch.from_numpy(pcm).cuda()
    res = []
    monkeypatch.setenv("VAMD_TEST_KNOBS", "1")   # (the margin is a test knob: ignored without this, vamd_knobs.h)
    for log2 in (None, "1", "-6"):
        if log2 is None:
            monkeypatch.delenv("VAMD_COUPLE_BAND_LOG2", raising=False)
        else:
            monkeypatch.setenv("VAMD_COUPLE_BAND_LOG2", log2)
        an = analyzer(name)
        assert ("VAMD_COUPLE_BAND_LOG2=%s" % (log2 or "unset:0")) in an.config_string()   # the knob really is in force
        outs = an.analyze(P, W=1, want=("iwork", "nonzero"))
        torch.cuda.synchronize()
        res.append({k: v.cpu().numpy() for k, v in outs.items()})
        an.close()
    assert np.abs(res[0]["iwork"]).max() > 20  # (the batch reaches magnitudes where steps are hit often)
    for r in res[1:]:
        assert np.array_equal(r["iwork"], res[0]["iwork"])
        assert np.array_equal(r["nonzero"], res[0]["nonzero"])


def test_mdct_forward_batch(torch_mod):
    torch = torch_mod
    chk = checker.Checker("44k_stereo_q4")
    an = analyzer("44k_stereo_q4")
    for W, nf in ((1, 4096), (0, 4096)):
        n = an.blocksizes[W]
        x = torch.rand((nf, n), device="cuda") - 0.5
        y = an.mdct_forward(W, x)
        torch.cuda.synchronize()
        xs, ys = x.cpu().numpy(), y.cpu().numpy()
        for i in list(range(0, nf, 97)) + [nf - 1]:
            assert np.array_equal(bits(ys[i]), bits(chk.mdct_forward(W, xs[i])))
    # empty batch is a no-op
    e = torch.empty((0, an.blocksizes[1]), device="cuda")
    assert an.mdct_forward(1, e).shape == (0, an.blocksizes[1] // 2)


def test_mdct_forward_every_size(torch_mod):
    """vamd_mdct_forward_batch at every transform size libvorbisenc sets up -- 256, 512, 1024, 2048, 4096 -- against the
    reference's mdct_forward.  (Round 5: the size-specialised fold took a form at n = 512 whose three regimes it assumed
    to change between a wave's trips; they change inside one there.  Nothing shipped reached it -- k_mdct_only is the
    C2 path, measured at 2048 -- and no test did either.)"""
    torch = torch_mod
    import vorbis_amd
    from oracle import ref
    seen = set()
    for ch, rate, q in ((2, 44100, 0.4), (1, 22050, 0.5), (2, 32000, 0.3), (2, 44100, -0.1), (1, 8000, 0.3), (2, 96000, 0.6)):
        e = ref.RefEncoder(ch, rate, q)
        an = vorbis_amd.Analyzer(e.pack_setup(), 0)
        for W in (0, 1):
            n = an.blocksizes[W]
            if n in seen:
                continue
            seen.add(n)
            x = torch.rand((300, n), device="cuda") - 0.5
            y = an.mdct_forward(W, x)
            torch.cuda.synchronize()
            xs, ys = x.cpu().numpy(), y.cpu().numpy()
            for i in range(0, 300, 7):
                assert np.array_equal(bits(ys[i]), bits(e.mdct_forward(W, xs[i]))), (n, i)
        an.close()
    assert seen >= {256, 512, 1024, 2048, 4096}, seen


def test_levels_and_workspace_vs_user_buffers(torch_mod):
    """LEVEL_TRANSFORM / LEVEL_PSY stop early; tensors kept in the internal workspace give the
    same downstream results as tensors written to caller buffers."""
    torch = torch_mod
    import vorbis_amd
    an = analyzer("44k_stereo_q4")
    pcm = torch.rand((64, 2, 2048), device="cuda") - 0.5
    full = an.analyze(pcm, want=ALL)
    lean = an.analyze(pcm, want=("iwork", "nonzero", "posts"))
    psy = an.analyze(pcm, level=vorbis_amd.LEVEL_PSY, want=("mdct_raw", "noise", "tone"))
    tr = an.analyze(pcm, level=vorbis_amd.LEVEL_TRANSFORM, want=("mdct_raw", "logfft", "logmdct", "local_ampmax"))
    torch.cuda.synchronize()
    for k in ("iwork", "nonzero", "posts"):
        assert torch.equal(full[k], lean[k])
    for k in ("mdct_raw", "noise", "tone"):
        assert torch.equal(full[k], psy[k])
    for k in ("mdct_raw", "logfft", "logmdct", "local_ampmax"):
        assert torch.equal(full[k], tr[k])


def test_host_block_api_matches_batch(torch_mod):
    torch = torch_mod
    an = analyzer("44k_stereo_q4")
    chk = checker.Checker("44k_stereo_q4")
    rng = np.random.default_rng(9)
    for W, lW, nW, bt in ((1, 1, 1, 1), (1, 0, 1, 0), (0, 0, 0, 1), (0, 0, 0, 0)):
        n = an.blocksizes[W]
        pcm = (rng.random((2, n), dtype=np.float32) - 0.5).astype(np.float32)
        o = an.analyze_block(pcm, lW, W, nW, bt, -50.0)
        ref = chk.tap_block(pcm, lW, W, nW, bt, -50.0)
        assert checker.compare_block(ref, o, an.posts[W], keys=("mdct", "logmask", "post_valid", "iwork", "nonzero"),
                                     verbose=True) == 0


def test_stream_mode_ampmax_chain(torch_mod):
    """vamd_analyze_stream reproduces the blockout->analysis ampmax recurrence (lib/block.c:626-628)."""
    torch = torch_mod
    chk = checker.Checker("44k_stereo_q4")
    an = analyzer("44k_stereo_q4")
    rng = np.random.default_rng(21)
    nb = 40
    gains = np.concatenate([np.full(10, 0.5), np.full(20, 0.001), np.full(10, 0.1)]).astype(np.float32)
    pcm = ((rng.random((nb, 2, 2048), dtype=np.float32) - 0.5) * 2 * gains[:, None, None]).astype(np.float32)
    outs, state = an.analyze_stream(torch.from_numpy(pcm).cuda(), -9999.0, want=("iwork", "ampmax_out", "posts", "nonzero"))
    torch.cuda.synchronize()
    amp = -9999.0
    enc = chk.enc
    got_amp = outs["ampmax_out"].cpu().numpy()
    iw = outs["iwork"].cpu().numpy()
    for b in range(nb):
        amp_in = enc.ampmax_decay(amp, 1)
        ref = chk.tap_block(pcm[b], ampmax_in=amp_in)
        amp = ref["ampmax_out"]
        assert np.float32(got_amp[b]) == np.float32(amp), b
        assert np.array_equal(iw[b], ref["iwork"]), b
    assert np.float32(state) == np.float32(amp)


def test_mixed_size_stream(torch_mod):
    """BASELINE config 5: a blockout-cut stream with interleaved short and long blocks; the GPU gets the
    blocks bucketed by size and the stream order, and must reproduce every block of the reference run
    (which includes the ampmax chain across size changes)."""
    from oracle import ref
    if not ref.available():
        pytest.skip("needs the reference build to cut a genuine stream")
    ch, rate, q = 2, 44100, 0.9
    rng = np.random.default_rng(404)
    frames = 44100 * 2
    t = np.arange(frames)
    gate = np.where((t % 11025) < 1102, 0.5, 0.0005).astype(np.float32)
    pcm = ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * gate).astype(np.float32)
    blocks = ref.RefEncoder(ch, rate, q).encode_stream(pcm)
    assert sum(1 for b in blocks if b["W"] == 0) > 20 and sum(1 for b in blocks if b["W"] == 1) > 20
    an = analyzer("44k_stereo_q9")
    res, state = an.analyze_stream_mixed(blocks, -9999.0, want=("mdct", "posts", "post_valid", "iwork", "nonzero", "ampmax_out"))
    chk = checker.Checker("44k_stereo_q9")
    for k, (b, g) in enumerate(zip(blocks, res)):
        r = chk.tap_block(b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])
        assert np.float32(g["ampmax_out"]) == np.float32(b["ampmax_out"]), k
        assert checker.compare_block(r, g, an.posts[b["W"]], keys=("mdct", "post_valid", "iwork", "nonzero"),
                                     verbose=True) == 0, k
    assert np.float32(state) == np.float32(blocks[-1]["ampmax_out"])


def test_full_size_properties(torch_mod):
    """BASELINE size (65 536 stereo blocks): size-independent properties instead of an oracle run.
    (1) determinism, (2) block k of a big batch == the same block analysed alone,
    (3) a batch made by tiling 64 oracle-verified blocks reproduces the verified outputs in
        every tile (checksum per tile), (4) MDCT linearity within fp32 rounding."""
    torch = torch_mod
    an = analyzer("44k_stereo_q4")
    chk = checker.Checker("44k_stereo_q4")
    nb = 65536
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    base = torch.rand((64, 2, 2048), generator=g, device="cuda") - 0.5
    want = ("mdct", "logmask", "iwork", "posts", "nonzero", "ampmax_out")
    small = an.analyze(base, want=want)
    torch.cuda.synchronize()
    b0 = base.cpu().numpy()
    for i in (0, 31, 63):
        ref = chk.tap_block(b0[i])
        assert checker.compare_block(ref, {k: v[i].cpu().numpy() for k, v in small.items()}, an.posts[1],
                                     keys=("mdct", "logmask", "iwork", "nonzero"), verbose=True) == 0
    big_in = base.repeat(nb // 64, 1, 1).contiguous()
    big = an.analyze(big_in, want=want)
    big2 = an.analyze(big_in, want=want)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(big[k], big2[k]), k                                   # (1)
        tiles = big[k].reshape((nb // 64, 64) + tuple(big[k].shape[1:]))
        assert bool((tiles == small[k].unsqueeze(0)).all()), k                  # (2),(3)
    x = torch.rand((4096, 2048), device="cuda") - 0.5
    y = torch.rand((4096, 2048), device="cuda") - 0.5
    lhs = an.mdct_forward(1, (x + y).contiguous())
    rhs = an.mdct_forward(1, x) + an.mdct_forward(1, y)
    assert float((lhs - rhs).abs().max()) < 2e-6                                # (4)


def test_argument_errors(torch_mod):
    import vorbis_amd
    an = analyzer("44k_stereo_q4")
    pcm = torch_mod.rand((4, 2, 2048), device="cuda")
    with pytest.raises(vorbis_amd.VamdError) as ei:
        an.analyze(pcm, blocktype=3)
    assert ei.value.code == -131  # OV_EINVAL
    with pytest.raises(vorbis_amd.VamdError):
        an.analyze(pcm, level=7, want=("mdct_raw",))


def test_context_on_second_device_while_first_is_current():
    """A vamd_ctx belongs to the device it was created on: every entry point switches to it and puts the caller's
    device back (vamd_create used to move the caller; later calls ran on whatever was current)."""
    import torch
    import vorbis_amd
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    torch.cuda.set_device(0)
    an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), device=1)
    assert torch.cuda.current_device() == 0
    rng = np.random.default_rng(3)
    pcm = (rng.random((4, 2, 2048), dtype=np.float32) - 0.5).astype(np.float32)
    outs = an.analyze(torch.from_numpy(pcm).to("cuda:1"))
    torch.cuda.synchronize(1)
    assert torch.cuda.current_device() == 0
    assert all(v.device.index == 1 for v in outs.values())
    chk = checker.Checker("44k_stereo_q4")
    for b in range(4):
        assert checker.compare_block(chk.tap_block(pcm[b]), {k: v[b].cpu().numpy() for k, v in outs.items()}, an.posts[1]) == 0
    with pytest.raises(ValueError):
        an.analyze(torch.from_numpy(pcm).to("cuda:0"))   # a tensor on the wrong device never reaches the kernels
    an.close()


def test_argument_checks_raise_value_errors():
    import torch
    import vorbis_amd
    an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob("44k_stereo_q4"), 0)
    good = torch.zeros((2, 2, 2048), device="cuda")
    with pytest.raises(ValueError):
        an.analyze(good.double())
    with pytest.raises(ValueError):
        an.analyze(good[:, :, ::2])
    with pytest.raises(ValueError):
        an.analyze(good.cpu())
    with pytest.raises(ValueError):
        an.analyze(good, outs={"mdct": torch.zeros((2, 2, 1024), dtype=torch.int32, device="cuda")})
    with pytest.raises(ValueError):
        an.analyze(good, lW=torch.zeros(3, dtype=torch.int32, device="cuda"))
    an.close()
