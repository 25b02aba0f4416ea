"""oracle/ref.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding to oracle/_ref/libvorbis_ref.so: the unmodified reference
libvorbis sources (compiled in place from /root/reference by oracle/Makefile)
plus oracle/ref_harness.c.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (vorbis_amd/) never does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libvorbis_ref.so")
# the same reference objects with lib/mapping0.c swapped for integration/mapping0_vamd.c (GPU path)
HYBRID_PATH = os.path.join(_HERE, "_ref", "libvorbis_hybrid.so")

BLOCKTYPE_IMPULSE = 0
BLOCKTYPE_PADDING = 1
BLOCKTYPE_TRANSITION = 0
BLOCKTYPE_LONG = 1

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_u8p = C.POINTER(C.c_ubyte)


class _Taps(C.Structure):
    _fields_ = [
        ("windowed", _f32p), ("mdct_raw", _f32p), ("fft_packed", _f32p), ("logfft", _f32p),
        ("logmdct", _f32p), ("noise", _f32p), ("tone", _f32p), ("logmask", _f32p), ("mdct", _f32p),
        ("posts", _i32p), ("post_valid", _i32p), ("ilogmask", _i32p), ("iwork", _i32p),
        ("nonzero", _i32p), ("local_ampmax", _f32p), ("ampmax_out", _f32p),
        ("packet", _u8p), ("packet_cap", C.c_long), ("packet_bytes", C.c_long),
        ("packet_matches_real", C.c_int),
        ("res_class", _i32p), ("res_class_cap", C.c_long), ("res_partvals", C.c_long),
        ("res_entries", C.POINTER(C.c_ushort)), ("res_entries_cap", C.c_long), ("res_count", C.c_long),
    ]


PACKETBLOBS = 15


class _MTaps(C.Structure):  # ref_mtaps (ref_harness.c): per candidate packet of a bitrate-managed encode
    _fields_ = [("posts", _i32p), ("post_valid", _i32p), ("ilogmask", _i32p), ("iwork", _i32p), ("nonzero", _i32p),
                ("packets", _u8p), ("packets_cap", C.c_long), ("packet_bytes", C.c_long * PACKETBLOBS),
                ("packets_match_real", C.c_int)]


class _BlockRec(C.Structure):
    _fields_ = [
        ("lW", C.c_int), ("W", C.c_int), ("nW", C.c_int), ("blocktype", C.c_int),
        ("ampmax_in", C.c_float), ("ampmax_out", C.c_float),
        ("pcm_offset", C.c_long), ("packet_offset", C.c_long), ("packet_bytes", C.c_long),
        ("granulepos", C.c_long), ("eos", C.c_long),
    ]


class _EnvState(C.Structure):  # ref_env_state (ref_harness.c): envelope_filter_state x channels + ve->stretch
    _fields_ = [("stretch", C.c_int), ("ampptr", C.c_int * 7 * 8), ("ampbuf", C.c_float * 17 * 7 * 8),
                ("nearptr", C.c_int * 8), ("nearDC", C.c_float * 15 * 8), ("nearDC_acc", C.c_float * 8),
                ("nearDC_partialacc", C.c_float * 8)]


def available():
    return os.path.exists(LIB_PATH)


def hybrid_available():
    return os.path.exists(HYBRID_PATH)


_libs = {}


def lib(hybrid=False):
    path = HYBRID_PATH if hybrid else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run `make -C oracle` (needs /root/reference)" % path)
        if hybrid:
            # the hybrid links libvorbis_amd.so and through it the HIP runtime; the tests also use torch, which
            # ships its own copy -- load torch's first so that the process ends up with one runtime
            import torch  # noqa: F401
        L = C.CDLL(path)
        L.ref_open.restype = C.c_void_p
        L.ref_open.argtypes = [C.c_int, C.c_long, C.c_float]
        L.ref_close.argtypes = [C.c_void_p]
        L.ref_pack_setup.restype = C.c_long
        L.ref_pack_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.ref_blocksize.argtypes = [C.c_void_p, C.c_int]
        L.ref_floor_posts.argtypes = [C.c_void_p, C.c_int]
        L.ref_apply_window.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int]
        L.ref_mdct_forward.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        L.ref_drft_forward.argtypes = [C.c_void_p, C.c_int, _f32p]
        L.ref_noisemask.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        L.ref_tonemask.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.c_float, C.c_float]
        L.ref_ampmax_decay.restype = C.c_float
        L.ref_ampmax_decay.argtypes = [C.c_void_p, C.c_float, C.c_int]
        L.ref_real_block.restype = C.c_long
        L.ref_real_block.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                     _u8p, C.c_long, _f32p]
        L.ref_tap_block.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.POINTER(_Taps)]
        L.ref_encode_stream.restype = C.c_long
        L.ref_encode_stream_ex.restype = C.c_long
        L.ref_encode_stream_ex.argtypes = [C.c_void_p, _f32p, C.c_long, C.c_long, C.c_int, C.POINTER(_BlockRec), C.c_long,
                                           _f32p, C.c_long, _u8p, C.c_long]
        L.ref_encode_stream.argtypes = [C.c_void_p, _f32p, C.c_long, C.POINTER(_BlockRec), C.c_long,
                                        _f32p, C.c_long, _u8p, C.c_long]
        L.ref_time_analysis.restype = C.c_double
        L.ref_time_analysis.argtypes = [C.c_void_p, _f32p, C.c_long, C.c_int]
        L.ref_time_dsp.restype = C.c_double
        L.ref_time_dsp.argtypes = [C.c_void_p, _f32p, C.c_long, C.c_int]
        L.ref_envelope_feed.restype = C.c_long
        L.ref_envelope_feed.argtypes = [C.c_void_p, _f32p, C.c_long]
        L.ref_envelope_get.restype = C.c_long
        L.ref_envelope_get.argtypes = [C.c_void_p, _f32p, C.c_long, C.POINTER(C.c_long), _i32p, C.c_long,
                                       C.POINTER(_EnvState)]
        L.ref_open_managed.restype = C.c_void_p
        L.ref_open_managed.argtypes = [C.c_int, C.c_long, C.c_long, C.c_long, C.c_long]
        L.ref_is_managed.argtypes = [C.c_void_p]
        L.ref_open_uncoupled.restype = C.c_void_p
        L.ref_open_uncoupled.argtypes = [C.c_int, C.c_long, C.c_float]
        L.ref_open_ctl.restype = C.c_void_p
        L.ref_open_ctl.argtypes = [C.c_int, C.c_long, C.c_float, C.c_double, C.c_double]
        L.ref_tap_block_managed.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.POINTER(_Taps), C.POINTER(_MTaps)]
        L.ref_matrix_case.restype = C.c_long
        L.ref_matrix_case.argtypes = [C.c_int, C.c_long, C.c_float, _f32p, C.c_int, _u8p, C.c_long,
                                      C.POINTER(C.c_long), C.c_long, _f32p, C.POINTER(C.c_long)]
        _libs[path] = L
    return _libs[path]


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _ip(a):
    return a.ctypes.data_as(_i32p)


class RefEncoder:
    """One reference encoder state (vorbis_info + vorbis_dsp_state + a vorbis_block)."""

    def __init__(self, channels=2, rate=44100, quality=0.4, hybrid=False, managed=None, coupled=True, lowpass_khz=None, iblock=None):
        """quality: libvorbisenc VBR quality; or managed=(max, nominal, min) bitrates for a
        bitrate-managed encoder (vorbis_encode_init), whose blocks carry 15 candidate packets.
        lowpass_khz / iblock: vorbis_encode_ctl OV_ECTL_LOWPASS_SET / OV_ECTL_IBLOCK_SET before setup_init."""
        self.L = lib(hybrid)
        if lowpass_khz is not None or iblock is not None:
            self.h = self.L.ref_open_ctl(channels, rate, quality, -1.0 if lowpass_khz is None else float(lowpass_khz),
                                         1.0 if iblock is None else float(iblock))
        elif not coupled:
            self.h = self.L.ref_open_uncoupled(channels, rate, quality)  # OV_ECTL_COUPLING_SET = 0
        elif managed is None:
            self.h = self.L.ref_open(channels, rate, quality)
        else:
            self.h = self.L.ref_open_managed(channels, rate, *[int(v) for v in managed])
        if not self.h:
            raise RuntimeError("vorbis_encode_init[_vbr] failed")
        self.channels, self.rate, self.quality, self.managed = channels, rate, quality, managed

    def close(self):
        if self.h:
            self.L.ref_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def blocksize(self, W):
        return self.L.ref_blocksize(self.h, W)

    def floor_posts(self, W):
        return self.L.ref_floor_posts(self.h, W)

    def pack_setup(self):
        need = self.L.ref_pack_setup(self.h, None, 0)
        if need < 0:
            raise RuntimeError("vamd_pack_setup failed: %d" % need)
        buf = np.zeros(need, dtype=np.uint8)
        got = self.L.ref_pack_setup(self.h, buf.ctypes.data_as(C.c_void_p), need)
        assert got == need
        return buf

    # ---- function-level taps ----
    def apply_window(self, d, lW, W, nW):
        d = np.ascontiguousarray(d, dtype=np.float32).copy()
        self.L.ref_apply_window(self.h, _fp(d), lW, W, nW)
        return d

    def mdct_forward(self, W, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.shape[-1] // 2, dtype=np.float32)
        self.L.ref_mdct_forward(self.h, W, _fp(x), _fp(out))
        return out

    def drft_forward(self, W, x):
        x = np.ascontiguousarray(x, dtype=np.float32).copy()
        self.L.ref_drft_forward(self.h, W, _fp(x))
        return x

    def noisemask(self, psy, logmdct):
        logmdct = np.ascontiguousarray(logmdct, dtype=np.float32)
        out = np.empty_like(logmdct)
        self.L.ref_noisemask(self.h, psy, _fp(logmdct), _fp(out))
        return out

    def tonemask(self, psy, logfft, global_ampmax, local_ampmax):
        logfft = np.ascontiguousarray(logfft, dtype=np.float32)
        out = np.empty_like(logfft)
        self.L.ref_tonemask(self.h, psy, _fp(logfft), _fp(out), global_ampmax, local_ampmax)
        return out

    def ampmax_decay(self, amp, W):
        return float(self.L.ref_ampmax_decay(self.h, amp, W))

    # ---- block-level ----
    def real_block(self, pcm, lW=1, W=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0):
        """The real vorbis_analysis() on one pre-cut block -> (packet bytes, ampmax_out)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        pkt = np.zeros(1 << 17, dtype=np.uint8)
        amp = C.c_float(0)
        nb = self.L.ref_real_block(self.h, _fp(pcm), lW, W, nW, blocktype, ampmax_in,
                                   pkt.ctypes.data_as(_u8p), pkt.size, C.byref(amp))
        if nb < 0:
            raise RuntimeError("vorbis_analysis failed: %d" % nb)
        return bytes(pkt[:nb]), amp.value

    def tap_block(self, pcm, lW=1, W=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0):
        """All mapping0_forward intermediates for one block pcm[ch][n] (dict of arrays)."""
        ch = self.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        n = self.blocksize(W)
        assert pcm.shape == (ch, n), pcm.shape
        n2 = n // 2
        o = {
            "windowed": np.empty((ch, n), np.float32), "mdct_raw": np.empty((ch, n2), np.float32),
            "fft_packed": np.empty((ch, n), np.float32), "logfft": np.empty((ch, n2), np.float32),
            "logmdct": np.empty((ch, n2), np.float32), "noise": np.empty((ch, n2), np.float32),
            "tone": np.empty((ch, n2), np.float32), "logmask": np.empty((ch, n2), np.float32),
            "mdct": np.empty((ch, n2), np.float32), "posts": np.zeros((ch, 65), np.int32),
            "post_valid": np.zeros(ch, np.int32), "ilogmask": np.empty((ch, n2), np.int32),
            "iwork": np.empty((ch, n2), np.int32), "nonzero": np.zeros(ch, np.int32),
            "local_ampmax": np.empty(ch, np.float32), "ampmax_out": np.empty(1, np.float32),
        }
        pkt = np.zeros(1 << 17, dtype=np.uint8)
        t = _Taps()
        for k, v in o.items():
            setattr(t, k, _fp(v) if v.dtype == np.float32 else _ip(v))
        t.packet = pkt.ctypes.data_as(_u8p)
        t.packet_cap = pkt.size
        rcls = np.zeros(2048, np.int32)
        rent = np.zeros(1 << 15, np.uint16)
        t.res_class, t.res_class_cap = _ip(rcls), rcls.size
        t.res_entries, t.res_entries_cap = rent.ctypes.data_as(C.POINTER(C.c_ushort)), rent.size
        ret = self.L.ref_tap_block(self.h, _fp(pcm), lW, W, nW, blocktype, ampmax_in, C.byref(t))
        if ret:
            raise RuntimeError("ref_tap_block failed: %d" % ret)
        o["packet"] = bytes(pkt[:t.packet_bytes])
        o["packet_matches_real"] = bool(t.packet_matches_real)
        o["ampmax_out"] = float(o["ampmax_out"][0])
        # residue back-end, submap after submap: classes per (partition, coded channel), and every codebook
        # entry in emission order
        assert t.res_count <= rent.size
        o["res_class"] = rcls[:t.res_partvals].copy()
        o["res_entries"] = rent[:t.res_count].copy()
        return o

    def tap_block_managed(self, pcm, lW=1, W=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0):
        """Managed-mode taps: the shared tensors of tap_block (mdct, logmask of select 1, ampmax) plus, per
        candidate packet k = 0..14, posts / post_valid / ilogmask / iwork / nonzero and the packet bytes."""
        ch = self.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        n = self.blocksize(W)
        assert pcm.shape == (ch, n), pcm.shape
        n2 = n // 2
        o = {"mdct_raw": np.empty((ch, n2), np.float32), "logfft": np.empty((ch, n2), np.float32),
             "logmdct": np.empty((ch, n2), np.float32), "noise": np.empty((ch, n2), np.float32),
             "tone": np.empty((ch, n2), np.float32), "logmask": np.empty((ch, n2), np.float32),
             "mdct": np.empty((ch, n2), np.float32), "local_ampmax": np.empty(ch, np.float32),
             "ampmax_out": np.empty(1, np.float32)}
        t = _Taps()
        for k, v in o.items():
            setattr(t, k, _fp(v))
        mo = {"posts": np.zeros((PACKETBLOBS, ch, 65), np.int32), "post_valid": np.zeros((PACKETBLOBS, ch), np.int32),
              "ilogmask": np.zeros((PACKETBLOBS, ch, n2), np.int32), "iwork": np.zeros((PACKETBLOBS, ch, n2), np.int32),
              "nonzero": np.zeros((PACKETBLOBS, ch), np.int32)}
        m = _MTaps()
        for k, v in mo.items():
            setattr(m, k, _ip(v))
        pk = np.zeros(1 << 20, np.uint8)
        m.packets, m.packets_cap = pk.ctypes.data_as(_u8p), pk.size
        ret = self.L.ref_tap_block_managed(self.h, _fp(pcm), lW, W, nW, blocktype, ampmax_in, C.byref(t), C.byref(m))
        if ret:
            raise RuntimeError("ref_tap_block_managed failed: %d" % ret)
        o["ampmax_out"] = float(o["ampmax_out"][0])
        for k, v in mo.items():
            o["m_" + k] = v
        sizes = [int(m.packet_bytes[k]) for k in range(PACKETBLOBS)]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        o["m_packets"] = [bytes(pk[offs[k]:offs[k + 1]]) for k in range(PACKETBLOBS)]
        o["packets_match_real"] = bool(m.packets_match_real)
        return o

    def encode_stream(self, pcm, max_blocks=1 << 16, write_frames=1024, tolerate=False, drain=0, jitter=0):
        """Run the whole application loop over planar pcm[ch][frames].  Consumes this
        encoder state.  Returns a list of dicts (lW,W,nW,blocktype,ampmax_in,ampmax_out,pcm,packet).
        write_frames: samples per vorbis_analysis_wrote() call (the example's 1024; the API takes any amount).
        tolerate: a failing vorbis_analysis() is recorded (`error` = its code, no packet) and the loop goes on.
        drain: after each write pull at most this many blocks (0 = all): blocks pile up while more samples arrive.
        jitter: a seed; every write is then a pseudo-random 1..write_frames samples, every pull 0..drain blocks."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        ch, frames = pcm.shape
        assert ch == self.channels
        recs = (_BlockRec * max_blocks)()
        pcm_cap = (frames * 2 + 8 * self.blocksize(1)) * ch * 2
        pcm_out = np.zeros(pcm_cap, np.float32)
        pk_cap = max(1 << 20, frames * ch)
        pk_out = np.zeros(pk_cap, np.uint8)
        self.L.ref_stream_set_drain(C.c_long(int(drain)))
        self.L.ref_stream_set_jitter(C.c_ulong(int(jitter)))
        nb = self.L.ref_encode_stream_ex(self.h, _fp(pcm), frames, int(write_frames), 1 if tolerate else 0, recs, max_blocks,
                                         _fp(pcm_out), pcm_cap, pk_out.ctypes.data_as(_u8p), pk_cap)
        self.L.ref_stream_set_drain(C.c_long(0))
        self.L.ref_stream_set_jitter(C.c_ulong(0))
        if nb < 0:
            raise RuntimeError("ref_encode_stream failed: %d" % nb)
        out = []
        for k in range(min(nb, max_blocks)):
            r = recs[k]
            n = self.blocksize(r.W)
            d = dict(lW=r.lW, W=r.W, nW=r.nW, blocktype=r.blocktype, ampmax_in=r.ampmax_in,
                     ampmax_out=r.ampmax_out, granulepos=int(r.granulepos), eos=int(r.eos))
            d["pcm"] = pcm_out[r.pcm_offset:r.pcm_offset + ch * n].reshape(ch, n).copy() if r.pcm_offset >= 0 else None
            d["packet"] = bytes(pk_out[r.packet_offset:r.packet_offset + r.packet_bytes]) if r.packet_offset >= 0 else None
            d["error"] = int(r.packet_bytes) if r.packet_bytes < 0 else 0
            out.append(d)
        return out

    def envelope_feed(self, pcm):
        """Append planar pcm[ch][frames] (vorbis_analysis_buffer/_wrote) and run the reference's own
        _ve_envelope_search over everything buffered.  Never shifts.  Returns a dict:
        steps (detector steps done since the start), pcm (the PCM ring as the detector saw it, including
        the centre padding / pre-extrapolation), marks [steps + 2], and the filter state in history form
        (stretch, near = the last 15 near-DC terms oldest first, amp [ch][7][17] oldest first,
        near_acc, near_partial)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        ch, frames = pcm.shape
        assert ch == self.channels
        steps = self.L.ref_envelope_feed(self.h, _fp(pcm), frames)
        if steps < 0:
            raise RuntimeError("ref_envelope_feed failed")
        n = C.c_long(0)
        self.L.ref_envelope_get(self.h, None, 0, C.byref(n), None, 0, None)
        seen = np.zeros((ch, n.value), np.float32)
        marks = np.zeros(steps + 2, np.int32)
        st = _EnvState()
        self.L.ref_envelope_get(self.h, _fp(seen), n.value, C.byref(n), _ip(marks), steps + 2, C.byref(st))
        amp = np.zeros((ch, 7, 17), np.float32)
        near = np.zeros((ch, 15), np.float32)
        for c in range(ch):
            ring = np.array(st.nearDC[c][:], np.float32)
            near[c] = np.roll(ring, -st.nearptr[c])          # nearptr = the oldest slot
            for b in range(7):
                a = np.array(st.ampbuf[c][b][:], np.float32)
                amp[c, b] = np.roll(a, -st.ampptr[c][b])     # ampptr = the oldest slot
        return dict(steps=int(steps), pcm=seen, marks=marks, stretch=int(st.stretch), near=near, amp=amp,
                    near_acc=np.array(st.nearDC_acc[:ch], np.float32),
                    near_partial=np.array(st.nearDC_partialacc[:ch], np.float32))

    def time_analysis(self, blocks, reps=1):
        blocks = np.ascontiguousarray(blocks, dtype=np.float32)
        return float(self.L.ref_time_analysis(self.h, _fp(blocks), blocks.shape[0], reps))

    def time_dsp(self, blocks, reps=1):
        blocks = np.ascontiguousarray(blocks, dtype=np.float32)
        return float(self.L.ref_time_dsp(self.h, _fp(blocks), blocks.shape[0], reps))


def matrix_case(ch, rate, q, data, hybrid=False):
    """One cell of the reference's own test grid (test/test.c:30-75) through ref_matrix_case (ref_harness.c):
    encode `data` (the same samples in every channel) as test/write_read.c does, decode the packets with the
    reference's vorbis_synthesis.  Returns (packets incl. the three headers, decoded channel 0)."""
    L = lib(hybrid)
    data = np.ascontiguousarray(data, dtype=np.float32)
    cap, maxp = 1 << 20, 256
    pk = np.zeros(cap, np.uint8)
    sizes = (C.c_long * maxp)()
    dec = np.full(len(data), 3.141, np.float32)   # set_data_in(data_in, .., 3.141), test/test.c:55
    total = C.c_long(0)
    n = L.ref_matrix_case(ch, rate, C.c_float(q), _fp(data), len(data), pk.ctypes.data_as(_u8p), cap, sizes, maxp,
                          _fp(dec), C.byref(total))
    if n < 0:
        raise RuntimeError("ref_matrix_case(%d, %d, %g) failed: %d" % (ch, rate, q, n))
    offs = np.concatenate([[0], np.cumsum([sizes[k] for k in range(n)])])
    return [bytes(pk[offs[k]:offs[k + 1]]) for k in range(n)], dec, int(total.value)
