"""oracle/port.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding to oracle/libvorbis_port.so, the from-scratch plain-C restatement of the hot path
(oracle/port/vorbis_port.c).  Same call shapes as oracle/ref.py so a test can swap one for the
other.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvorbis_port.so")
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)


class _Taps(C.Structure):
    _fields_ = [(k, _f32p) for k in ("windowed", "mdct_raw", "fft_packed", "logfft", "logmdct", "noise", "tone",
                                     "logmask", "mdct")] + \
               [(k, _i32p) for k in ("posts", "post_valid", "ilogmask", "iwork", "nonzero")] + \
               [("local_ampmax", _f32p), ("ampmax_out", _f32p)] + \
               [("res_class", _i32p), ("res_class_cap", C.c_long), ("res_partvals", C.c_long),
                ("res_entries", C.POINTER(C.c_ushort)), ("res_entries_cap", C.c_long), ("res_count", C.c_long)]


class _PortEnvFilter(C.Structure):  # port_env_filter == envelope_filter_state (lib/envelope.h:34-45)
    _fields_ = [("ampbuf", C.c_float * 17), ("ampptr", C.c_int), ("nearDC", C.c_float * 15),
                ("nearDC_acc", C.c_float), ("nearDC_partialacc", C.c_float), ("nearptr", C.c_int)]


class PortEnvState(C.Structure):  # all-zero = start of stream
    _fields_ = [("stretch", C.c_int), ("f", _PortEnvFilter * 7 * 2)]


class _MTaps(C.Structure):  # port_mtaps: per candidate packet of a bitrate-managed block
    _fields_ = [(k, _i32p) for k in ("posts", "post_valid", "ilogmask", "iwork", "nonzero")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE, "port"])
        L = C.CDLL(LIB_PATH)
        L.port_open.restype = C.c_void_p
        L.port_open.argtypes = [C.c_void_p, C.c_size_t]
        L.port_close.argtypes = [C.c_void_p]
        for f in ("port_channels", "port_blocksize", "port_floor_posts"):
            getattr(L, f).argtypes = [C.c_void_p] + ([C.c_int] if f != "port_channels" else [])
        L.port_apply_window.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int]
        L.port_mdct_forward.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        L.port_drft_forward.argtypes = [C.c_void_p, C.c_int, _f32p]
        L.port_noisemask.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        L.port_tonemask.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.c_float, C.c_float]
        L.port_ampmax_decay.restype = C.c_float
        L.port_ampmax_decay.argtypes = [C.c_void_p, C.c_float, C.c_int]
        L.port_tap_block.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.POINTER(_Taps)]
        L.port_time_dsp.restype = C.c_double
        L.port_time_dsp.argtypes = [C.c_void_p, _f32p, C.c_long, C.c_int]
        L.port_tap_block_managed.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                             C.POINTER(_Taps), C.POINTER(_MTaps)]
        L.port_envelope_steps.argtypes = [C.c_void_p, C.POINTER(PortEnvState), _f32p, C.c_long, C.c_long, C.c_void_p]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


class PortEncoder:
    def __init__(self, setup_blob):
        self.L = lib()
        blob = np.ascontiguousarray(setup_blob, dtype=np.uint8)
        self.h = self.L.port_open(blob.ctypes.data_as(C.c_void_p), blob.size)
        if not self.h:
            raise RuntimeError("port_open: bad setup blob")
        self.channels = self.L.port_channels(self.h)

    def close(self):
        if self.h:
            self.L.port_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def blocksize(self, W):
        return self.L.port_blocksize(self.h, W)

    def floor_posts(self, W):
        return self.L.port_floor_posts(self.h, W)

    def apply_window(self, d, lW, W, nW):
        d = np.ascontiguousarray(d, dtype=np.float32).copy()
        self.L.port_apply_window(self.h, _fp(d), lW, W, nW)
        return d

    def mdct_forward(self, W, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.shape[-1] // 2, dtype=np.float32)
        self.L.port_mdct_forward(self.h, W, _fp(x), _fp(out))
        return out

    def drft_forward(self, W, x):
        x = np.ascontiguousarray(x, dtype=np.float32).copy()
        self.L.port_drft_forward(self.h, W, _fp(x))
        return x

    def noisemask(self, psy, logmdct):
        logmdct = np.ascontiguousarray(logmdct, dtype=np.float32)
        out = np.empty_like(logmdct)
        self.L.port_noisemask(self.h, psy, _fp(logmdct), _fp(out))
        return out

    def tonemask(self, psy, logfft, global_ampmax, local_ampmax):
        logfft = np.ascontiguousarray(logfft, dtype=np.float32)
        out = np.empty_like(logfft)
        self.L.port_tonemask(self.h, psy, _fp(logfft), _fp(out), global_ampmax, local_ampmax)
        return out

    def ampmax_decay(self, amp, W):
        return float(self.L.port_ampmax_decay(self.h, amp, W))

    def tap_block(self, pcm, lW=1, W=1, nW=1, blocktype=1, ampmax_in=-9999.0):
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
        t = _Taps()
        for k, v in o.items():
            setattr(t, k, v.ctypes.data_as(_f32p if v.dtype == np.float32 else _i32p))
        rcls = np.zeros(2048, np.int32)
        rent = np.zeros(1 << 15, np.uint16)
        t.res_class, t.res_class_cap = rcls.ctypes.data_as(_i32p), rcls.size
        t.res_entries, t.res_entries_cap = rent.ctypes.data_as(C.POINTER(C.c_ushort)), rent.size
        r = self.L.port_tap_block(self.h, _fp(pcm), lW, W, nW, blocktype, ampmax_in, C.byref(t))
        if r:
            raise RuntimeError("port_tap_block failed: %d" % r)
        o["ampmax_out"] = float(o["ampmax_out"][0])
        o["res_class"] = rcls[:t.res_partvals].copy()
        o["res_entries"] = rent[:t.res_count].copy()
        return o

    def tap_block_managed(self, pcm, lW=1, W=1, nW=1, blocktype=1, ampmax_in=-9999.0):
        """All 15 candidate packets' floors / residues of one block (same keys as RefEncoder.tap_block_managed)."""
        ch = self.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        n = self.blocksize(W)
        n2 = n // 2
        o = {"mdct": np.empty((ch, n2), np.float32), "logmask": np.empty((ch, n2), np.float32),
             "ampmax_out": np.empty(1, np.float32)}
        t = _Taps()
        for k, v in o.items():
            setattr(t, k, v.ctypes.data_as(_f32p))
        mo = {"posts": np.zeros((15, ch, 65), np.int32), "post_valid": np.zeros((15, ch), np.int32),
              "ilogmask": np.zeros((15, ch, n2), np.int32), "iwork": np.zeros((15, ch, n2), np.int32),
              "nonzero": np.zeros((15, ch), np.int32)}
        m = _MTaps()
        for k, v in mo.items():
            setattr(m, k, v.ctypes.data_as(_i32p))
        r = self.L.port_tap_block_managed(self.h, _fp(pcm), lW, W, nW, blocktype, ampmax_in, C.byref(t), C.byref(m))
        if r:
            raise RuntimeError("port_tap_block_managed failed: %d" % r)
        o["ampmax_out"] = float(o["ampmax_out"][0])
        for k, v in mo.items():
            o["m_" + k] = v
        return o

    def envelope_steps(self, pcm, nsteps, state=None):
        """The step loop of _ve_envelope_search over planar pcm[ch][len]; returns (flags uint8[nsteps], state)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if state is None:
            state = PortEnvState()
        ret = np.zeros(nsteps, np.uint8)
        r = self.L.port_envelope_steps(self.h, C.byref(state), _fp(pcm), pcm.shape[1], nsteps,
                                       ret.ctypes.data_as(C.c_void_p))
        if r:
            raise RuntimeError("port_envelope_steps failed: %d" % r)
        return ret, state

    def time_dsp(self, blocks, reps=1):
        blocks = np.ascontiguousarray(blocks, dtype=np.float32)
        return float(self.L.port_time_dsp(self.h, _fp(blocks), blocks.shape[0], reps))
