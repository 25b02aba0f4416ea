"""tests/soak_lib.py -- random-signal soak of the GPU batch path against the reference's real vorbis_analysis():
many random blocks of eight signal kinds (noise over 90 dB of level, sines, impulses, clipping, DC ramps, silent
channels, s16 decaying harmonics, anti-phase pairs) with random window flags, block types and incoming ampmax,
twelve configurations (1-8 channels, 22-96 kHz, q -0.1 .. 0.9, coupled and not), both block sizes; every packet and
ampmax compared with oracle/_ref.  Then bitrate-managed blocks: all fifteen candidate packets each.
Test infrastructure: driven by tests/test_gpu_soak.py (-m gpu) and tools/soak.py."""
import time

import numpy as np
import torch

import vorbis_amd
from oracle import ref

CONFIGS = [(2, 44100, 0.4, True), (2, 44100, 0.1, True), (2, 44100, 0.9, True), (2, 44100, -0.1, True), (1, 44100, 0.3, True),
           (6, 44100, 0.3, True), (6, 44100, 0.7, True), (2, 44100, 0.5, False), (4, 44100, 0.2, True), (2, 22050, 0.3, True),
           (2, 96000, 0.6, True), (8, 48000, 0.5, True)]


def signals(rng, nb, ch, n):
    x = np.zeros((nb, ch, n), np.float32)
    t = np.arange(n, dtype=np.float64)
    for k in range(nb):
        kind = k % 8
        amp = 10.0 ** rng.uniform(-4.5, 0)
        if k % 64 == 63:
            amp *= 8.0    # over full scale now and then: a spectral maximum above 0 dB is clamped (lib/mapping0.c:345)
        if kind == 0:
            x[k] = (rng.random((ch, n)) - 0.5) * 2 * amp
        elif kind == 1:   # a few sines, channels correlated
            s = sum(np.sin(2 * np.pi * rng.uniform(0.0005, 0.45) * t + rng.uniform(0, 6)) for _ in range(3)) / 3
            x[k] = amp * s[None, :] * rng.uniform(0.2, 1.0, (ch, 1))
        elif kind == 2:   # impulse on a quiet floor
            x[k] = (rng.random((ch, n)) - 0.5) * 1e-4
            x[k, :, rng.integers(0, n)] = amp
        elif kind == 3:   # clipped noise
            x[k] = np.clip((rng.random((ch, n)) - 0.5) * 8 * amp, -amp, amp)
        elif kind == 4:   # DC plus ramp
            x[k] = amp * (0.3 + np.linspace(-1, 1, n))[None, :] * rng.uniform(-1, 1, (ch, 1))
        elif kind == 5:   # some channels silent
            x[k] = (rng.random((ch, n)) - 0.5) * 2 * amp
            x[k, rng.integers(0, ch)] = 0
        elif kind == 6:   # s16-quantised music-like: decaying harmonics
            f0 = rng.uniform(0.001, 0.02)
            s = sum(np.sin(2 * np.pi * f0 * h * t) / h for h in range(1, 12)) * np.exp(-t / rng.uniform(200, 4000))
            x[k] = np.round(amp * 0.5 * s[None, :] * rng.uniform(0.5, 1.0, (ch, 1)) * 32767) / 32768
        else:             # anti-phase stereo pairs
            b = (rng.random(n) - 0.5) * 2 * amp
            for c in range(ch):
                x[k, c] = b * (-1 if c & 1 else 1) * rng.uniform(0.8, 1.0)
    return x



def run(NB=300, managed=True, log=print):
    """Returns (blocks compared, mismatching blocks)."""
    bad = total = 0
    t0 = time.time()
    for ch, rate, q, coupled in CONFIGS:
        e = ref.RefEncoder(ch, rate, q, coupled=coupled)
        an = vorbis_amd.Analyzer(e.pack_setup(), 0)
        rng = np.random.default_rng(hash((ch, rate, int(q * 10))) & 0xffff)
        for W in (1, 0):
            n = e.blocksize(W)
            nb = NB if W else NB // 3
            x = signals(rng, nb, ch, n)
            lW = rng.integers(0, 2, nb).astype(np.int32) * W
            nW = rng.integers(0, 2, nb).astype(np.int32) * W
            bt = (rng.integers(0, 2, nb)).astype(np.int32)
            amp_in = np.where(rng.random(nb) < 0.5, -9999.0, rng.uniform(-60, 0, nb)).astype(np.float32)
            o = an.analyze(torch.from_numpy(x).cuda(), W=W, lW=lW, nW=nW, blocktype=bt, ampmax_in=amp_in,
                           want=("ampmax_out", "packets", "packet_bits"))
            torch.cuda.synchronize()
            rows, bits, amps = o["packets"].cpu().numpy(), o["packet_bits"].cpu().numpy(), o["ampmax_out"].cpu().numpy()
            for k in range(nb):
                a = e.tap_block(x[k], int(lW[k]), W, int(nW[k]), int(bt[k]), float(amp_in[k]))
                ok = a["packet_matches_real"] and vorbis_amd.packet_bytes(rows[k], bits[k]) == a["packet"] and \
                    np.float32(amps[k]) == np.float32(a["ampmax_out"])
                total += 1
                if not ok:
                    bad += 1
                    log("MISMATCH", (ch, rate, q, coupled), "W", W, "block", k, "kind", k % 8)
        an.close()
        log("%d ch %d Hz q %.1f coupled=%s done, %d blocks so far, %d mismatches, %.0f s" % (ch, rate, q, coupled, total, bad, time.time() - t0))
    if not managed:
        return total, bad
    # bitrate-managed: all fifteen candidate packets of every block
    for ch, rates in ((2, (-1, 128000, -1)), (2, (-1, 64000, -1)), (6, (-1, 256000, -1)), (1, (64000, 48000, 32000))):
        e = ref.RefEncoder(ch, 44100, managed=rates)
        an = vorbis_amd.Analyzer(e.pack_setup(), 0)
        rng = np.random.default_rng(ch * 7 + rates[1] // 1000)
        for W in (1, 0):
            n = e.blocksize(W)
            nb = max(8, NB // 10) if W else max(8, NB // 30)
            x = signals(rng, nb, ch, n)
            lW = rng.integers(0, 2, nb).astype(np.int32) * W
            nW = rng.integers(0, 2, nb).astype(np.int32) * W
            o = an.analyze_managed(torch.from_numpy(x).cuda(), W=W, lW=lW, nW=nW, blocktype=1 if W else 0, packets=True)
            torch.cuda.synchronize()
            rows, bits = o["m_packets"].cpu().numpy(), o["m_packet_bits"].cpu().numpy()
            for k in range(nb):
                a = e.tap_block_managed(x[k], int(lW[k]), W, int(nW[k]), 1 if W else 0)
                ok = a["packets_match_real"] and [vorbis_amd.packet_bytes(rows[k, j], bits[k, j]) for j in range(15)] == a["m_packets"]
                total += 1
                if not ok:
                    bad += 1
                    log("MISMATCH managed", ch, rates, "W", W, "block", k, "kind", k % 8)
        an.close()
        log("managed %d ch %s done, %d blocks so far, %d mismatches, %.0f s" % (ch, rates, total, bad, time.time() - t0))
    return total, bad
