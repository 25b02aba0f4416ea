"""tests/soak_lib.py -- random-signal soak of the GPU batch path against the reference's real vorbis_analysis():
many random blocks of fourteen signal kinds (noise over 90 dB of level, sines, impulses, clipping, DC ramps, silent
channels, s16 decaying harmonics, anti-phase pairs; and the edges of the input domain: denormal-level noise, signed
zeros around single denormals, noise 30-60 dB over full scale, impulses 86 dB over full scale on a denormal floor,
and -- round 5: the domain's integer edge is now the arithmetic's own, not a +60 dB margin -- noise 60-90 dB and a
sine 70-85 dB over full scale, quantised values in the thousands)
with random window flags, block types and incoming ampmax,
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
        kind = k % NKINDS
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
        elif kind == 7:   # anti-phase stereo pairs
            b = (rng.random(n) - 0.5) * 2 * amp
            for c in range(ch):
                x[k, c] = b * (-1 if c & 1 else 1) * rng.uniform(0.8, 1.0)
        # ---- the edges of the input domain (include/vorbis_amd.h): finite, so the reference's result is defined
        # and must be met bit for bit.  x86 computes with denormals (no FTZ/DAZ); so must every GPU instruction.
        elif kind == 8:   # denormal-level noise: every sample, and most of the spectrum, below FLT_MIN
            x[k] = ((rng.random((ch, n)) - 0.5) * 2 * 10.0 ** rng.uniform(-44.5, -37.5)).astype(np.float32)
        elif kind == 9:   # signed zeros with single denormals / tiny normals in them
            x[k] = np.where(rng.random((ch, n)) < 0.5, np.float32(-0.0), np.float32(0.0))
            for _ in range(int(rng.integers(0, 4))):
                x[k, rng.integers(0, ch), rng.integers(0, n)] = np.float32(rng.choice([1e-45, -1e-45, 3e-39, -1.2e-38, 1e-30]))
        elif kind == 10:  # far over full scale, inside the domain (white noise of amplitude A peaks ~20 dB under A in a
            # 2048-block, ~11 dB under it in a 256-block: below +50 dB)
            x[k] = ((rng.random((ch, n)) - 0.5) * 2 * 10.0 ** rng.uniform(1.5, 3.0)).astype(np.float32)
        elif kind == 11:  # an impulse 86 dB over full scale on a denormal floor (flat spectrum at 4A/n: +50 dB in a 256-block)
            x[k] = ((rng.random((ch, n)) - 0.5) * 1e-39).astype(np.float32)
            x[k, :, rng.integers(0, n)] = np.float32(rng.choice([-2e4, 2e4]))
        # ---- between the old +60 dB line and the integer edge (vamd_quant_limit, ~ +85 dB): quantised values in the
        # thousands, every floor at its 0 dB ceiling -- defined arithmetic in the reference, so bit-exact here
        elif kind == 12:  # noise 60 .. 90 dB over full scale (spectrum ~ 0.3 A: values up to ~ 9 000)
            x[k] = ((rng.random((ch, n)) - 0.5) * 2 * 10.0 ** rng.uniform(3.0, 4.5)).astype(np.float32)
        else:             # a sine 70 .. 85 dB over full scale in one or all channels, the rest ordinary noise
            x[k] = ((rng.random((ch, n)) - 0.5) * 0.2).astype(np.float32)
            s = np.sin(2 * np.pi * rng.uniform(0.002, 0.45) * t + rng.uniform(0, 6)) * 10.0 ** rng.uniform(3.5, 4.25)
            if rng.random() < 0.5:
                x[k, rng.integers(0, ch)] += s.astype(np.float32)
            else:
                x[k] += (s[None, :] * rng.uniform(0.3, 1.0, (ch, 1))).astype(np.float32)
    return x


NKINDS = 14



def run(NB=300, managed=True, log=print):
    """Returns (blocks compared, mismatching blocks)."""
    bad = total = beyond = 0
    t0 = time.time()
    for ch, rate, q, coupled in CONFIGS:
        e = ref.RefEncoder(ch, rate, q, coupled=coupled)
        an = vorbis_amd.Analyzer(e.pack_setup(), 0)
        rng = np.random.default_rng(hash((ch, rate, int(q * 10))) & 0xffff)
        an_q = [0, 0]
        for W in (1, 0):
            n = e.blocksize(W)
            nb = NB if W else NB // 3
            x = signals(rng, nb, ch, n)
            lW = rng.integers(0, 2, nb).astype(np.int32) * W
            nW = rng.integers(0, 2, nb).astype(np.int32) * W
            bt = (rng.integers(0, 2, nb)).astype(np.int32)
            amp_in = np.where(rng.random(nb) < 0.5, -9999.0, rng.uniform(-60, 0, nb)).astype(np.float32)
            o = an.analyze(torch.from_numpy(x).cuda(), W=W, lW=lW, nW=nW, blocktype=bt, ampmax_in=amp_in,
                           want=("ampmax_out", "packets", "packet_bits", "status"))
            torch.cuda.synchronize()
            st = o["status"].cpu().numpy()
            flagged = an.input_status()
            if flagged != (int((st != 0).sum()), 0):   # the two doors agree
                bad += 1
                log("FLAGGED count differs", (ch, rate, q, coupled), "W", W, flagged, int((st != 0).sum()))
            an_q[W] = min(an.quant_limit(W, c)[0] for c in range(ch))
            rows, bits, amps = o["packets"].cpu().numpy(), o["packet_bits"].cpu().numpy(), o["ampmax_out"].cpu().numpy()
            for k in range(nb):
                a = e.tap_block(x[k], int(lW[k]), W, int(nW[k]), int(bt[k]), float(amp_in[k]))
                # every soak signal is finite; a block is outside the domain exactly where the reference's own quantised
                # values pass the setup's bound (kinds 12 / 13 get there now and then) -- channel by channel
                want_st = an.beyond_quant_limit(W, a["iwork"])
                total += 1
                if not np.array_equal(st[k], want_st):
                    bad += 1
                    log("STATUS differs", (ch, rate, q, coupled), "W", W, "block", k, "kind", k % NKINDS, st[k].tolist(), want_st.tolist())
                    continue
                if want_st.any():
                    beyond += 1
                    ok = np.float32(amps[k]) == np.float32(a["ampmax_out"])   # (the ampmax chain is carried over such a block)
                    if not ok:
                        bad += 1
                        log("MISMATCH ampmax of a block beyond the bound", (ch, rate, q, coupled), "W", W, "block", k)
                    continue
                ok = a["packet_matches_real"] and vorbis_amd.packet_bytes(rows[k], bits[k]) == a["packet"] and \
                    np.float32(amps[k]) == np.float32(a["ampmax_out"])
                if not ok:
                    bad += 1
                    log("MISMATCH", (ch, rate, q, coupled), "W", W, "block", k, "kind", k % NKINDS)
        an.close()
        log("%d ch %d Hz q %.1f coupled=%s done, %d blocks so far (%d of them beyond the integer bound %d / %d and reported), %d mismatches, %.0f s"
            % (ch, rate, q, coupled, total, beyond, an_q[0], an_q[1], bad, time.time() - t0))
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
            o = an.analyze_managed(torch.from_numpy(x).cuda(), W=W, lW=lW, nW=nW, blocktype=1 if W else 0, packets=True,
                                   want=("status",))
            torch.cuda.synchronize()
            st = o["status"].cpu().numpy()
            an.input_status()
            rows, bits = o["m_packets"].cpu().numpy(), o["m_packet_bits"].cpu().numpy()
            for k in range(nb):
                a = e.tap_block_managed(x[k], int(lW[k]), W, int(nW[k]), 1 if W else 0)
                # (a channel-block is beyond the bound when ANY of its fifteen candidates' values is)
                want_st = np.bitwise_or.reduce([an.beyond_quant_limit(W, a["m_iwork"][j]) for j in range(15)])
                total += 1
                if not np.array_equal(st[k], want_st):
                    bad += 1
                    log("STATUS differs (managed)", ch, rates, "W", W, "block", k, "kind", k % NKINDS, st[k].tolist(), want_st.tolist())
                    continue
                if want_st.any():
                    continue
                ok = a["packets_match_real"] and [vorbis_amd.packet_bytes(rows[k, j], bits[k, j]) for j in range(15)] == a["m_packets"]
                if not ok:
                    bad += 1
                    log("MISMATCH managed", ch, rates, "W", W, "block", k, "kind", k % NKINDS)
        an.close()
        log("managed %d ch %s done, %d blocks so far, %d mismatches, %.0f s" % (ch, rates, total, bad, time.time() - t0))
    return total, bad


# ---- outside the input domain -------------------------------------------------------------------------------------
HOSTILE = ("nan", "+inf", "-inf", "1e30", "fltmax", "nan_everywhere", "nan_in_zeroed_window", "sine+94dB")
NONFINITE, RANGE = vorbis_amd.api.STATUS_NONFINITE, vorbis_amd.api.STATUS_RANGE


def hostile_batch(rng, nb, ch, n, W):
    """nb blocks of ordinary noise; every third one gets a hostile sample in ONE channel.  Returns
    (pcm, lW, nW, expected non-finite status bit [nb][ch], blocks whose hostile sample is finite)."""
    x = ((rng.random((nb, ch, n)) - 0.5) * 2 * 0.3).astype(np.float32)
    lW = np.ones(nb, np.int32) * W
    nW = np.ones(nb, np.int32) * W
    want = np.zeros((nb, ch), np.uint8)
    loud = []
    for k in range(0, nb, 3):
        kind = HOSTILE[(k // 3) % len(HOSTILE)]
        c = int(rng.integers(0, ch))
        pos = int(rng.integers(n // 4 + 8, 3 * n // 4 - 8))   # inside every window shape's non-zero part
        flagged = NONFINITE
        if kind == "nan":
            x[k, c, pos] = np.nan
        elif kind == "+inf":
            x[k, c, pos] = np.inf
        elif kind == "-inf":
            x[k, c, pos] = -np.inf
        elif kind == "1e30":       # finite, but the reference's own fp32 power spectrum overflows to Inf: the same class
            x[k, c, pos] = 1e30
        elif kind == "fltmax":
            x[k, c, pos] = -3.4028235e38
        elif kind == "nan_everywhere":
            x[k, c, :] = np.nan
        elif kind == "sine+94dB":   # finite arithmetic, but quantised values of ~35 000: past every setup's integer bound
            x[k, c] = (5e4 * np.sin(0.3 * np.arange(n))).astype(np.float32)   # (and short of 46 341, lib/psy.c:985)
            flagged = 0
            loud.append(k)
        else:
            # a long block after a short one: _vorbis_apply_window ZEROES [0, n/4 - bs0/4) instead of multiplying
            # (lib/window.c:2117-2118), so a NaN there never enters the arithmetic -- the block is inside the
            # domain, its result is the reference's, and it must NOT be flagged
            if W:
                lW[k] = 0
                x[k, c, 3] = np.nan
                flagged = 0
            else:
                x[k, c, pos] = np.nan
        want[k, c] = flagged
    return x, lW, nW, want, loud


def run_hostile(nb=48, log=print):
    """Blocks outside the input domain are reported (status, vamd_input_status), never crash or hang, and leave
    every other block of their batch bit-exact.  Returns (checks made, failures)."""
    from vorbis_amd import VamdError
    checks = bad = 0
    for ch, rate, q in ((2, 44100, 0.4), (2, 44100, 0.1), (6, 44100, 0.3), (1, 22050, 0.5)):
        e = ref.RefEncoder(ch, rate, q)
        an = vorbis_amd.Analyzer(e.pack_setup(), 0)
        rng = np.random.default_rng(ch * 1000 + int(q * 10))
        for W in (1, 0):
            n = e.blocksize(W)
            x, lW, nW, want, loud = hostile_batch(rng, nb, ch, n, W)
            o = an.analyze(torch.from_numpy(x).cuda(), W=W, lW=lW, nW=nW, blocktype=1 if W else 0,
                           want=("ampmax_out", "packets", "packet_bits", "status"))
            torch.cuda.synchronize()
            st = o["status"].cpu().numpy()
            rows, bits, amps = o["packets"].cpu().numpy(), o["packet_bits"].cpu().numpy(), o["ampmax_out"].cpu().numpy()
            checks += 1
            if not np.array_equal(st & NONFINITE, want):     # the non-finite bit: exactly the channel that holds the sample
                bad += 1
                log("HOSTILE non-finite status differs", (ch, rate, q), "W", W, np.argwhere((st & NONFINITE) != want)[:8].tolist())
            counted = an.input_status()
            checks += 1
            if counted != (int((st != 0).sum()), 0) or an.last_input_code != vorbis_amd.VAMD_ENONFINITE or an.input_status() != (0, 0):
                bad += 1
                log("HOSTILE count differs", (ch, rate, q), "W", W, counted, int((st != 0).sum()), an.last_input_code)
            beyond = set()   # the finite blocks the reference's own values put past a bound (the loud ones, if any)
            for k in range(nb):
                if want[k].any():
                    continue   # (deterministic but unspecified; the range bit may come on top where saturated values spread by coupling)
                a = e.tap_block(x[k], int(lW[k]), W, int(nW[k]), 1 if W else 0, -9999.0)
                # the range bit: exactly the channels whose quantised values (the reference's own, defined up to here)
                # pass the setup's bound
                want_r = an.beyond_quant_limit(W, a["iwork"])
                checks += 1
                if want_r.any():
                    beyond.add(k)
                if not np.array_equal(st[k], want_r) or (want_r.any() and k not in loud):
                    bad += 1
                    log("HOSTILE range status differs", (ch, rate, q), "W", W, "block", k, st[k].tolist(), want_r.tolist())
                checks += 1
                if np.float32(amps[k]) != np.float32(a["ampmax_out"]):   # (delivered for a block beyond the bound too)
                    bad += 1
                    log("HOSTILE ampmax differs", (ch, rate, q), "W", W, "block", k)
                if want_r.any():
                    continue
                checks += 1
                if vorbis_amd.packet_bytes(rows[k], bits[k]) != a["packet"]:
                    bad += 1
                    log("HOSTILE clean block differs", (ch, rate, q), "W", W, "block", k)
            checks += 1
            if (ch, W) == (2, 1) and not beyond:      # (the loud kind is there to cross the bound: in stereo long blocks it must)
                bad += 1
                log("HOSTILE no block crossed the integer bound", (ch, rate, q), "W", W)
            # the host-pointer entry point says so itself, and which of the two it was
            for k in [0, 1, 3] + loud[:1]:
                checks += 1
                code = 0
                try:
                    an.analyze_block(x[k], int(lW[k]), W, int(nW[k]), 1 if W else 0, -9999.0)
                except VamdError as err:
                    code = err.code
                expect = vorbis_amd.VAMD_ENONFINITE if want[k].any() else (vorbis_amd.VAMD_EDOMAIN if k in beyond else 0)
                if code != expect:
                    bad += 1
                    log("HOSTILE analyze_block verdict wrong", (ch, rate, q), "W", W, "block", k, code, expect)
            an.input_status()   # (the host calls' blocks count too: start the next batch from zero)
        # the detector: clean steps are the reference's, a NaN is an error, and the state survives for the next stream
        steps = 40
        stream = ((rng.random((ch, 64 * (steps + 2))) - 0.5) * 0.2).astype(np.float32)
        checks += 1
        try:
            poisoned = stream.copy()
            poisoned[ch - 1, 777] = np.nan
            an.envelope_search(poisoned, steps)
            bad += 1
            log("HOSTILE detector accepted a NaN", (ch, rate, q))
        except VamdError as err:
            if err.code != vorbis_amd.VAMD_ENONFINITE:
                bad += 1
                log("HOSTILE detector: wrong error", err)
        an.close()
        log("hostile %d ch %d Hz q %.1f done, %d checks, %d failures" % (ch, rate, q, checks, bad))
    return checks, bad
