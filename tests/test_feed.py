"""GPU suite: the host-fed farm (vamd_feed, include/vorbis_amd.h) -- whole streams in from host memory as 16-bit
interleaved samples, finished packets back -- against the reference encoder run over the same samples the way
examples/encoder_example.c:179-236 runs it (x / 32768.f, 1024 frames per vorbis_analysis_wrote(), closed with
vorbis_analysis_wrote(v, 0)): packet count, every packet's bytes, granule positions, size class and the
end-of-stream flag, from a stream's first block (behind the reference's backward LPC extrapolation,
lib/block.c:417-458) to its last (inside its forward one, :474-512)."""
import numpy as np
import pytest

from tests import checker

pytestmark = pytest.mark.gpu


def s16_streams(rng, ch, frames, kinds):
    out = []
    t = np.arange(frames)
    for kind in kinds:
        if kind == "noise":
            x = (rng.random((frames, ch)) - 0.5) * 0.8
        elif kind == "gated":
            gate = np.where((t % 9000) < 700, 0.5, 0.0005)[:, None]
            x = (rng.random((frames, ch)) - 0.5) * 2 * gate
        elif kind == "sine":
            x = 0.6 * np.sin(2 * np.pi * 440.0 / 44100.0 * t)[:, None] * np.ones((1, ch)) + (rng.random((frames, ch)) - 0.5) * 1e-3
        elif kind == "clicks":
            x = (rng.random((frames, ch)) - 0.5) * 0.002
            x[::4001] = 0.9
        elif kind == "silence":
            x = np.zeros((frames, ch))
        else:
            raise ValueError(kind)
        out.append(np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16))
    return np.stack(out)


def reference_packets(setup, pcm_s16):
    """The reference's application loop over one stream's samples [frames, ch] int16."""
    from oracle import ref
    ch, rate, q = checker.SETUPS[setup]
    planar = np.ascontiguousarray((pcm_s16.astype(np.float32) / np.float32(32768.0)).T)
    return ref.RefEncoder(ch, rate, q).encode_stream(planar)


def compare(setup, pcm, got):
    bad = []
    for s in range(pcm.shape[0]):
        want = reference_packets(setup, pcm[s])
        if len(want) != len(got[s]):
            bad.append("stream %d: %d packets, the reference %d" % (s, len(got[s]), len(want)))
            continue
        for k, (w, g) in enumerate(zip(want, got[s])):
            data, gp, W, eos = g
            if data != w["packet"] or gp != w["granulepos"] or W != w["W"] or eos != w["eos"]:
                bad.append("stream %d packet %d/%d: bytes %s granulepos %d/%d W %d/%d eos %d/%d" % (
                    s, k, len(want), "equal" if data == w["packet"] else "DIFFER", gp, w["granulepos"], W, w["W"], eos, w["eos"]))
                break
    return bad


@pytest.mark.parametrize("setup", ["44k_stereo_q4", "44k_stereo_q9", "44k_mono_q5"])
def test_whole_streams_from_host_s16_match_the_reference(setup):
    import vorbis_amd
    from oracle import ref
    if not ref.available():
        pytest.skip("needs the reference build")
    ch = checker.SETUPS[setup][0]
    rng = np.random.default_rng(2026)
    frames = 30000
    pcm = s16_streams(rng, ch, frames, ["noise", "gated", "sine", "clicks", "silence", "gated"])
    feed = vorbis_amd.Feed(vorbis_amd.default_setup_blob(setup), lanes_per_device=2, max_streams=8, max_frames=frames)
    got = feed.encode(pcm)
    feed.close()
    assert all(len(g) > 20 for g in got)
    bad = compare(setup, pcm, got)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("frames", [1, 20, 33, 100, 1023, 1500, 2048, 2049, 2500, 3071, 3072, 3073, 4097, 5000, 7777])
def test_short_streams(frames):
    """Streams too short for the pre-extrapolation to run before they close, for a full long block, for LPC at all."""
    import vorbis_amd
    from oracle import ref
    if not ref.available():
        pytest.skip("needs the reference build")
    setup = "44k_stereo_q4"
    rng = np.random.default_rng(frames)
    pcm = s16_streams(rng, 2, frames, ["noise", "gated", "sine"])
    feed = vorbis_amd.Feed(vorbis_amd.default_setup_blob(setup), lanes_per_device=1, max_streams=4, max_frames=8192)
    got = feed.encode(pcm)
    feed.close()
    bad = compare(setup, pcm, got)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("as_float", [False, True])
def test_groups_in_flight(as_float):
    """Three lanes, seven groups of different sizes kept in flight; float input gives what 16-bit input gives."""
    import vorbis_amd
    from oracle import ref
    if not ref.available():
        pytest.skip("needs the reference build")
    setup = "44k_stereo_q4"
    rng = np.random.default_rng(5)
    feed = vorbis_amd.Feed(vorbis_amd.default_setup_blob(setup), lanes_per_device=3, max_streams=6, max_frames=20000,
                           fmt=vorbis_amd.FEED_F32 if as_float else vorbis_amd.FEED_S16)
    groups = [s16_streams(rng, 2, fr, kinds) for fr, kinds in
              [(20000, ["noise", "gated"]), (9000, ["sine"]), (15000, ["clicks", "noise", "gated", "sine"]), (20000, ["gated"] * 6),
               (5000, ["noise"]), (12345, ["gated", "noise"]), (20000, ["sine", "silence"])]]
    pending, results = [], {}
    for gi, pcm in enumerate(groups):
        if len(pending) == 3:
            g0, slot = pending.pop(0)
            results[g0] = feed.packets(slot)
            feed.release(slot)
        slot, buf = feed.buffer(2)
        flat = pcm.reshape(-1)
        buf[:flat.size] = (flat.astype(np.float32) / np.float32(32768.0)) if as_float else flat
        feed.wrote(slot, pcm.shape[0], pcm.shape[1])
        pending.append((gi, slot))
    for g0, slot in pending:
        results[g0] = feed.packets(slot)
        feed.release(slot)
    feed.close()
    for gi, pcm in enumerate(groups):
        r = results[gi]
        got = []
        for s in range(pcm.shape[0]):
            row = []
            for k in range(int(r["stream_start"][s]), int(r["stream_start"][s + 1])):
                bits, o = int(r["bits"][k]), int(r["offset"][k])
                row.append((bytes(r["bytes"][o:o + (bits + 7) // 8]), int(r["granulepos"][k]), int(r["info"][k]) & 1, (int(r["info"][k]) >> 1) & 1))
            got.append(row)
        bad = compare(setup, pcm, got)
        assert not bad, "group %d\n" % gi + "\n".join(bad)


@pytest.mark.parametrize("setup", ["44k_stereo_q4", "44k_stereo_q9"])
def test_streams_of_unequal_length_in_one_group(setup):
    """vamd_feed_wrote_v: eleven streams from 1 to 40 000 frames back to back in one group -- every stream gets its own LPC
    ends, its own share of the detector's steps and its own walk, and emits what the reference emits for it alone."""
    import vorbis_amd
    from oracle import ref
    if not ref.available():
        pytest.skip("needs the reference build")
    rng = np.random.default_rng(31)
    lengths = [40000, 1, 33, 2049, 17000, 3072, 39999, 700, 25000, 4097, 12345]
    kinds = ["gated", "noise", "sine", "clicks", "noise", "gated", "sine", "gated", "clicks", "noise", "gated"]
    parts = [s16_streams(rng, 2, n, [k])[0] for n, k in zip(lengths, kinds)]
    feed = vorbis_amd.Feed(vorbis_amd.default_setup_blob(setup), lanes_per_device=2, max_streams=16, max_frames=40000)
    got = feed.encode(parts)
    again = feed.encode(parts[::-1])[::-1]          # (the same streams in another order, on the other lane)
    feed.close()
    bad = []
    for s, (pcm, g) in enumerate(zip(parts, got)):
        bad += compare(setup, pcm[None], [g])
        assert again[s] == g, "stream %d depends on its place in the group" % s
    assert not bad, "\n".join(bad)


def test_feed_argument_errors():
    import vorbis_amd
    feed = vorbis_amd.Feed(vorbis_amd.default_setup_blob("44k_stereo_q4"), lanes_per_device=1, max_streams=2, max_frames=4096)
    slot, _ = feed.buffer(2)
    with pytest.raises(vorbis_amd.VamdError):
        feed.wrote(slot, 3, 100)          # more streams than the feed holds
    with pytest.raises(vorbis_amd.VamdError):
        feed.wrote(slot, 1, 5000)         # longer than the feed holds
    with pytest.raises(vorbis_amd.VamdError):
        feed.buffer(2)                    # the only lane is out and nothing is in flight: waiting would be for ever
    with pytest.raises(vorbis_amd.VamdError):
        feed.packets(slot)                # nothing was written
    feed.release(slot)
    feed.close()
