"""ctypes wrapper of tests/emul/libvamd_emul.so (host-compiled kernel bodies; test build only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
LIB = os.path.join(_HERE, "libvamd_emul.so")
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)

POSTS_STRIDE = 32
RES_CLASS_STRIDE = 512


class _Taps(C.Structure):
    _fields_ = [(k, _f32p) for k in ("mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct")] + \
               [(k, _i32p) for k in ("posts", "post_valid", "ilogmask", "iwork", "nonzero")] + \
               [("local_ampmax", _f32p), ("ampmax_out", _f32p)] + \
               [("res_class", _i32p), ("res_entries", C.POINTER(C.c_ushort)), ("res_count", _i32p)] + \
               [("packet", C.c_void_p), ("packet_bits", _i32p)]


def packet_bytes(row, bits):
    """The packet as oggpack_get_buffer()/oggpack_bytes() would hand it over: (bits+7)/8 bytes, the
    unused high bits of the last one zero."""
    nbytes = (bits + 7) // 8
    assert nbytes <= row.size, "packet longer than its row"
    return row[:nbytes].tobytes()


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))] + \
           [os.path.join(_ROOT, "vorbis_amd", "csrc", f)
                                                 for f in os.listdir(os.path.join(_ROOT, "vorbis_amd", "csrc"))
                                                 if f.endswith(".h")] + \
           [os.path.join(_ROOT, "include", f) for f in os.listdir(os.path.join(_ROOT, "include"))]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unknown-pragmas",
           "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_ROOT, "vorbis_amd", "csrc"), "-I" + _HERE,
           "-shared", "-o", LIB, os.path.join(_HERE, "emul.cpp")]
    subprocess.check_call(cmd)
    return LIB


class _MTaps(C.Structure):
    _fields_ = [(k, _i32p) for k in ("posts", "post_valid", "iwork", "nonzero")] + \
               [("packets", C.c_void_p), ("packet_bits", _i32p)]


class Emul:
    def __init__(self, blob):
        self.L = C.CDLL(build())
        self.L.emul_open.restype = C.c_void_p
        self.L.emul_open.argtypes = [C.c_void_p, C.c_size_t]
        self.L.emul_close.argtypes = [C.c_void_p]
        self.L.emul_mdct_forward.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        self.L.emul_analyze_block.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                              C.POINTER(_Taps)]
        self.L.emul_analyze_block_managed.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                                      C.POINTER(_Taps), C.POINTER(_MTaps)]
        self.L.emul_residue_capacity.argtypes = [C.c_void_p, C.c_int]
        self.L.emul_packet_capacity.argtypes = [C.c_void_p, C.c_int]
        self.L.emul_submaps.argtypes = [C.c_void_p, C.c_int]
        self.L.emul_residue_offset.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.emul_envelope_search.argtypes = [C.c_void_p, _f32p, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
        self.L.emul_plan_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
        self.L.emul_lpc_head.argtypes = [_f32p, C.c_int, C.c_int]
        self.L.emul_lpc_head.restype = None
        self.L.emul_lpc_tail.argtypes = [_f32p, C.c_long, C.c_long, C.c_int, C.c_int]
        self.L.emul_lpc_tail.restype = None
        self.L.emul_chase_compare.argtypes = [_f32p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.h = self.L.emul_open(blob.ctypes.data_as(C.c_void_p), blob.size)
        if not self.h:
            raise RuntimeError("emul_open failed")
        hdr = np.frombuffer(blob.tobytes()[:40], dtype=np.int32)
        self.channels, self.bs = int(hdr[4]), (int(hdr[6]), int(hdr[7]))

    def mdct_forward(self, W, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.size // 2, np.float32)
        self.L.emul_mdct_forward(self.h, W, x.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
        return out

    def analyze_block(self, pcm, lW=1, W=1, nW=1, blocktype=1, ampmax_in=-9999.0):
        ch, n = self.channels, self.bs[W]
        n2 = n // 2
        pcm = np.ascontiguousarray(pcm, np.float32)
        assert pcm.shape == (ch, n)
        o = {k: np.zeros((ch, n2), np.float32) for k in ("mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct")}
        o["posts"] = np.zeros((ch, POSTS_STRIDE), np.int32)
        o["post_valid"] = np.zeros(ch, np.int32)
        o["ilogmask"] = np.zeros((ch, n2), np.int32)
        o["iwork"] = np.zeros((ch, n2), np.int32)
        o["nonzero"] = np.zeros(ch, np.int32)
        o["local_ampmax"] = np.zeros(ch, np.float32)
        o["ampmax_out"] = np.zeros(1, np.float32)
        t = _Taps()
        for k, v in o.items():
            setattr(t, k, v.ctypes.data_as(_f32p if v.dtype == np.float32 else _i32p))
        cap = self.L.emul_residue_capacity(self.h, W)
        if cap > 0:
            S = self.L.emul_submaps(self.h, W)
            rcls, rent, rcnt = np.zeros(S * RES_CLASS_STRIDE, np.int32), np.zeros(cap, np.uint16), np.zeros(2 * S, np.int32)
            t.res_class, t.res_count = rcls.ctypes.data_as(_i32p), rcnt.ctypes.data_as(_i32p)
            t.res_entries = rent.ctypes.data_as(C.POINTER(C.c_ushort))
            pk = np.full(self.L.emul_packet_capacity(self.h, W), 0xAA, np.uint8)
            pbits = np.zeros(1, np.int32)
            t.packet, t.packet_bits = pk.ctypes.data, pbits.ctypes.data_as(_i32p)
        r = self.L.emul_analyze_block(self.h, pcm.ctypes.data_as(_f32p), lW, W, nW, blocktype, ampmax_in, C.byref(t))
        assert r == 0
        o["ampmax_out"] = float(o["ampmax_out"][0])
        if cap > 0:
            # all submaps' classes / entries one after the other (the order the reference emits them in)
            offs = [self.L.emul_residue_offset(self.h, W, sm) for sm in range(S)]
            o["res_class"] = np.concatenate([rcls[sm * RES_CLASS_STRIDE:sm * RES_CLASS_STRIDE + rcnt[2 * sm]] for sm in range(S)])
            o["res_entries"] = np.concatenate([rent[offs[sm]:offs[sm] + rcnt[2 * sm + 1]] for sm in range(S)])
            o["packet_bits"] = int(pbits[0])
            o["packet"] = packet_bytes(pk, int(pbits[0]))
        return o

    def analyze_block_managed(self, pcm, lW=1, W=1, nW=1, blocktype=1, ampmax_in=-9999.0):
        """All 15 candidate packets of a bitrate-managed block (keys as RefEncoder.tap_block_managed)."""
        ch, n = self.channels, self.bs[W]
        n2 = n // 2
        pcm = np.ascontiguousarray(pcm, np.float32)
        o = {"mdct": np.zeros((ch, n2), np.float32), "logmask": np.zeros((ch, n2), np.float32),
             "ampmax_out": np.zeros(1, np.float32)}
        t = _Taps()
        for k, v in o.items():
            setattr(t, k, v.ctypes.data_as(_f32p))
        mo = {"posts": np.zeros((15, ch, POSTS_STRIDE), np.int32), "post_valid": np.zeros((15, ch), np.int32),
              "iwork": np.zeros((15, ch, n2), np.int32), "nonzero": np.zeros((15, ch), np.int32)}
        m = _MTaps()
        for k, v in mo.items():
            setattr(m, k, v.ctypes.data_as(_i32p))
        pcap = self.L.emul_packet_capacity(self.h, W)
        if pcap > 0:
            pk, pbits = np.full((15, pcap), 0xAA, np.uint8), np.zeros(15, np.int32)
            m.packets, m.packet_bits = pk.ctypes.data, pbits.ctypes.data_as(_i32p)
        r = self.L.emul_analyze_block_managed(self.h, pcm.ctypes.data_as(_f32p), lW, W, nW, blocktype, ampmax_in,
                                              C.byref(t), C.byref(m))
        assert r == 0
        o["ampmax_out"] = float(o["ampmax_out"][0])
        for k, v in mo.items():
            o["m_" + k] = v
        if pcap > 0:
            o["m_packets"] = [packet_bytes(pk[k], int(pbits[k])) for k in range(15)]
        return o

    def envelope_search(self, pcm, nsteps, state):
        """Same contract as Analyzer.envelope_search; `state` is a vorbis_amd.EnvelopeState."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        ret = np.zeros(nsteps, np.uint8)
        r = self.L.emul_envelope_search(self.h, pcm.ctypes.data_as(_f32p), pcm.shape[1], nsteps, C.byref(state),
                                        ret.ctypes.data_as(C.c_void_p))
        assert r == 0
        return ret

    def plan_stream(self, flags, nsamples, maxblocks=4096, eof=0, with_pending=False):
        """blockout's decisions for one stream from its detector flags (k_blockout.h compiled for the host).
        Returns (kind[n], begin[n]): kind = W | lW << 1 | nW << 2 | blocktype << 3.  eof: the stream's end (BlockoutP::eof);
        with_pending: also the centre of the block the walk stopped in front of."""
        flags = np.ascontiguousarray(flags, np.uint8)
        kind = np.zeros(maxblocks, np.int32)
        begin = np.zeros(maxblocks, np.int32)
        pending = C.c_long(0)
        n = self.L.emul_plan_stream(self.h, flags.ctypes.data_as(C.c_void_p), C.c_long(len(flags)), C.c_long(nsamples),
                                    maxblocks, kind.ctypes.data_as(C.c_void_p), begin.ctypes.data_as(C.c_void_p), C.c_long(eof),
                                    C.byref(pending))
        return (kind[:n], begin[:n], int(pending.value)) if with_pending else (kind[:n], begin[:n])

    def whole_stream(self, pcm, write_frames=1024):
        """vamd_plan_streams_whole's sequence for ONE stream, every piece the product's own code compiled for the host: pcm
        [ch][frames] -> (the stream's buffer with both LPC ends filled, kind[], begin[]).  The detector flags come from
        envelope_search (k_envelope.h)."""
        import vorbis_amd
        pcm = np.ascontiguousarray(pcm, np.float32)
        ch, frames = pcm.shape
        bs1, head, pad = self.bs[1], self.bs[1] // 2, 3 * self.bs[1]
        buf = np.zeros((ch, head + frames + pad + 64), np.float32)
        buf[:, head:head + frames] = pcm
        n_head = min(frames, (bs1 // write_frames + 1) * write_frames)
        for c in range(ch):
            self.L.emul_lpc_head(buf[c].ctypes.data_as(_f32p), head, n_head)
        steps1 = max(0, (head + frames) // 64 - 4)
        st = vorbis_amd.EnvelopeState()
        flags1 = self.envelope_search(buf[:, :(steps1 - 1) * 64 + 128], steps1, st) if steps1 else np.zeros(0, np.uint8)
        _, _, pending = self.plan_stream(flags1, head + frames, with_pending=True)
        for c in range(ch):
            self.L.emul_lpc_tail(buf[c].ctypes.data_as(_f32p), head + frames, pending - bs1 // 2, bs1, pad)
        steps_all = (head + frames + pad) // 64 - 4
        n2 = steps_all - steps1
        flags2 = self.envelope_search(buf[:, steps1 * 64:steps1 * 64 + (n2 - 1) * 64 + 128], n2, st)
        kind, begin = self.plan_stream(np.concatenate([flags1, flags2]), head + frames + pad, eof=head + frames)
        return buf, kind, begin

    def fit_segments_mismatches(self):
        """accumulate_fit's static work list against the reference's loops (emul_fit_segments_check)."""
        self.L.emul_fit_segments_check.restype = C.c_long
        self.L.emul_fit_segments_check.argtypes = [C.c_void_p]
        return int(self.L.emul_fit_segments_check(self.h))

    def div_magic_mismatches(self):
        """div_magic() against integer division over the floor line walks' domain (emul_div_magic_check)."""
        self.L.emul_div_magic_check.restype = C.c_long
        return int(self.L.emul_div_magic_check())

    def couple_estimate_mismatches(self):
        """chan_bin_sure / couple_bin_sure against the exact forms (emul_couple_estimate_check): (mismatches, unsure ppm)."""
        self.L.emul_couple_estimate_check.restype = C.c_long
        ppm = C.c_long(0)
        bad = int(self.L.emul_couple_estimate_check(C.byref(ppm)))
        return bad, int(ppm.value)

    def quant_energy_mismatches(self):
        """quant_energy() against the reference's fp64 expression (emul_quant_energy_check)."""
        self.L.emul_quant_energy_check.restype = C.c_long
        return int(self.L.emul_quant_energy_check())

    def chase_compare(self, seeds, linesper):
        """seed_chase's stack walk serially and in 64 verified chunks (k_tone.h chase_chunk) over the same seed
        lines: (lists agree, chunks accepted, survivors, repair rounds)."""
        n = len(seeds)
        buf = np.full((n + 31) & ~15, -9999.0, np.float32)
        buf[:n] = seeds
        acc, ns, rd = C.c_int(0), C.c_int(0), C.c_int(0)
        same = self.L.emul_chase_compare(buf.ctypes.data_as(_f32p), linesper, n, C.byref(acc), C.byref(ns), C.byref(rd))
        return bool(same), bool(acc.value), ns.value, rd.value
