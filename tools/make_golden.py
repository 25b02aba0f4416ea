"""Regenerate tests/golden/*.npz from the reference itself (oracle/_ref = the unmodified
libvorbis sources compiled in place).  Needs /root/reference, so it only runs in the build
container; the fixtures it writes travel with the repo.

Per setup: blocks cut by the reference's own vorbis_analysis_blockout from a gated-noise stream
(long / short / transition windows, impulse / padding block types, genuine ampmax chain), two
white-noise long blocks with a fresh ampmax, one block of digital silence and one of a pure tone;
for every block all mapping0_forward taps plus the packet bytes the real vorbis_analysis() emits.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests.checker import SETUPS, SURROUND  # noqa: E402

TAPS = ("windowed", "mdct_raw", "fft_packed", "logfft", "logmdct", "noise", "tone", "logmask", "mdct", "posts",
        "post_valid", "ilogmask", "iwork", "nonzero", "local_ampmax", "res_class", "res_entries")


# six channels: the decisions and the packet only (the float taps would be 6 x the size)
TAPS_SURROUND = ("posts", "post_valid", "iwork", "nonzero", "local_ampmax", "res_class", "res_entries")


def pick(blocks):
    """A small, varied subset of stream blocks."""
    seen, out = set(), []
    for b in blocks:
        key = (b["W"], b["lW"] if b["W"] else 0, b["nW"] if b["W"] else 0, b["blocktype"])
        if key not in seen:
            seen.add(key)
            out.append(b)
    return out


def main():
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for name, (ch, rate, q) in list(SETUPS.items()) + list(SURROUND.items()):
        rng = np.random.default_rng(20240925)
        frames = 44100 * 2
        t = np.arange(frames)
        gate = np.where((t % 11025) < 1102, 0.5, 0.0005).astype(np.float32)
        stream = ((rng.random((ch, frames), dtype=np.float32) - 0.5) * 2 * gate).astype(np.float32)
        if ch > 2:   # correlated fronts, a quiet LFE: lossless and point coupling both occur
            stream[1] = 0.8 * stream[0] + 0.2 * stream[1]
            stream[5] *= 0.05
        blocks = pick(ref.RefEncoder(ch, rate, q).encode_stream(stream))
        e = ref.RefEncoder(ch, rate, q)
        n = e.blocksize(1)
        extra = []
        for amp in (0.5, 0.02):
            extra.append(dict(lW=1, W=1, nW=1, blocktype=1, ampmax_in=-9999.0,
                              pcm=((rng.random((ch, n), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)))
        extra.append(dict(lW=1, W=1, nW=1, blocktype=1, ampmax_in=-9999.0, pcm=np.zeros((ch, n), np.float32)))
        tone = (0.4 * np.sin(2 * np.pi * 1000.0 / rate * np.arange(n))).astype(np.float32)
        extra.append(dict(lW=1, W=1, nW=1, blocktype=1, ampmax_in=-30.0,
                          pcm=np.stack([tone * (1.0 - 0.3 * c) for c in range(ch)]).astype(np.float32)))
        rec = {}
        allb = blocks + extra
        for i, b in enumerate(allb):
            r = e.tap_block(b["pcm"], b["lW"], b["W"], b["nW"], b["blocktype"], b["ampmax_in"])
            assert r["packet_matches_real"]
            if b.get("packet") is not None:
                assert r["packet"] == b["packet"], "tap harness disagrees with the stream run"
            rec["b%d_desc" % i] = np.array([b["lW"], b["W"], b["nW"], b["blocktype"]], np.int32)
            rec["b%d_ampmax" % i] = np.array([b["ampmax_in"], r["ampmax_out"]], np.float32)
            rec["b%d_pcm" % i] = b["pcm"]
            rec["b%d_packet" % i] = np.frombuffer(r["packet"], np.uint8)
            for k in (TAPS if ch <= 2 else TAPS_SURROUND):
                rec["b%d_%s" % (i, k)] = r[k]
        rec["nblocks"] = np.array([len(allb)], np.int32)
        rec["posts"] = np.array([e.floor_posts(0), e.floor_posts(1)], np.int32)
        # function-level vectors: mdct_forward / drft_forward on raw random frames, both sizes
        for W in (0, 1):
            x = (rng.random(e.blocksize(W), dtype=np.float32) - 0.5).astype(np.float32)
            rec["mdct%d_in" % W] = x
            rec["mdct%d_out" % W] = e.mdct_forward(W, x)
            rec["drft%d_out" % W] = e.drft_forward(W, x)
        path = os.path.join(ROOT, "tests", "golden", "blocks_%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, len(allb), "blocks ->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
