"""ctypes mirror of include/vorbis_amd.h.  Names and argument meaning follow the C ABI, which in
turn follows libvorbis (mapping0_forward's variables and OV_* return codes)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

VAMD_OK, VAMD_EFAULT, VAMD_EIMPL, VAMD_EINVAL, VAMD_EVERSION, VAMD_EDOMAIN, VAMD_ENONFINITE = 0, -129, -130, -131, -134, -140, -141
ABI_VERSION = 9             # VAMD_ABI_VERSION of the header this mirror was written against
QUANT_LIMIT_SQUARE, QUANT_LIMIT_INT = 46340, 0x7fffff80
STATUS_RANGE, STATUS_NONFINITE = 1, 2   # bits of the `status` output (vorbis_amd.h, "Input domain")
LEVEL_TRANSFORM, LEVEL_PSY, LEVEL_FULL = 1, 2, 3
POSTS_STRIDE = 32
BLOCKTYPE_IMPULSE, BLOCKTYPE_PADDING, BLOCKTYPE_TRANSITION, BLOCKTYPE_LONG = 0, 1, 0, 1

# every symbol include/vorbis_amd.h declares
EXPORTED_SYMBOLS = ["vamd_create_abi", "vamd_encode_blocks", "vamd_clock_probe", "vamd_quant_limit", "vamd_config_string", "vamd_destroy", "vamd_last_error", "vamd_set_stream", "vamd_reserve",
                    "vamd_channels", "vamd_blocksize", "vamd_posts", "vamd_mdct_forward_batch",
                    "vamd_analyze_batch", "vamd_analyze_stream", "vamd_analyze_block", "vamd_profile",
                    "vamd_stage_ms", "vamd_debug_cycles", "vamd_analyze_stream_mixed", "vamd_envelope_search_batch",
                    "vamd_envelope_search", "vamd_envelope_geometry", "vamd_residue_capacity", "vamd_analyze_block_res", "vamd_analyze_batch_managed", "vamd_analyze_block_managed",
                    "vamd_packet_capacity", "vamd_encode_block", "vamd_submaps", "vamd_residue_offset", "vamd_analyze_streams_mixed",
                    "vamd_plan_streams", "vamd_gather_blocks", "vamd_plan_fetch",
                    "vamd_batcher_create", "vamd_batcher_destroy", "vamd_batcher_attach", "vamd_batcher_detach",
                    "vamd_batcher_encode_block", "vamd_batcher_last_error", "vamd_batcher_stats", "vamd_batcher_context", "vamd_batcher_report",
                    "vamd_input_status", "vamd_calib_copy", "vamd_abi_version", "vamd_plan_streams_whole", "vamd_plan_streams_whole_v", "vamd_feed_wrote_v", "vamd_device_count", "vamd_batcher_create_multi",
                    "vamd_feed_create", "vamd_feed_destroy", "vamd_feed_lanes", "vamd_feed_device", "vamd_feed_buffer", "vamd_feed_wrote",
                    "vamd_feed_packets", "vamd_feed_release", "vamd_feed_last_error"]
PACKETBLOBS = 15

_vp = C.c_void_p


class _Plan(C.Structure):
    _fields_ = [("nstreams", C.c_int64), ("nblocks", C.c_int64 * 2), ("lW", _vp * 2), ("nW", _vp * 2), ("blocktype", _vp * 2),
                ("src", _vp * 2), ("order", _vp), ("stream_start", _vp)]


class _FeedResult(C.Structure):  # vamd_feed_result
    _fields_ = [("nstreams", C.c_int64), ("nblocks", C.c_int64), ("stream_start", _vp), ("offset", _vp), ("bits", _vp),
                ("granulepos", _vp), ("info", _vp), ("bytes", _vp), ("total_bytes", C.c_int64), ("upload_ms", C.c_double),
                ("device_ms", C.c_double), ("total_ms", C.c_double)]


class _Desc(C.Structure):
    _fields_ = [("W", C.c_int), ("nblocks", C.c_long), ("lW", _vp), ("nW", _vp), ("blocktype", _vp),
                ("ampmax_in", _vp), ("uniform_lW", C.c_int), ("uniform_nW", C.c_int),
                ("uniform_blocktype", C.c_int), ("uniform_ampmax_in", C.c_float)]


_IO_FIELDS = ["pcm", "mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct", "posts", "post_valid",
              "ilogmask", "iwork", "nonzero", "local_ampmax", "ampmax_out", "res_class", "res_entries", "res_count"]
RES_CLASS_STRIDE = 512
MAX_CH = 8


class _IO(C.Structure):
    _fields_ = [(k, _vp) for k in _IO_FIELDS] + [("packets", _vp), ("packet_bits", _vp), ("packet_stride", C.c_int64),
                                                 ("status", _vp), ("pcm_src", _vp), ("pcm_channel_stride", C.c_int64)]


class _MIO(C.Structure):  # vamd_managed_io
    _fields_ = [(k, _vp) for k in ("posts", "post_valid", "iwork", "nonzero", "res_class", "res_entries", "res_count",
                                   "packets", "packet_bits")] + [("packet_stride", C.c_int64)]


def packet_bytes(row, bits):
    """A packet as oggpack_get_buffer() / oggpack_bytes() hand it over: the first (bits+7)/8 bytes of its row."""
    nbytes = (int(bits) + 7) // 8
    if nbytes > len(row):
        raise ValueError("packet (%d bits) was cut off at its row length %d" % (bits, len(row)))
    return bytes(bytearray(row[:nbytes]))


class VamdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vorbis_amd error %d: %s" % (code, msg))
        self.code = code


def library_path():
    return os.path.join(_HERE, "libvorbis_amd.so")


_lib = None


def load_library():
    """Load libvorbis_amd.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("vorbis_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    # The library and torch must share ONE HIP runtime (device pointers and streams cross between them): torch
    # ships its own libamdhip64, so it is loaded first and the dlopen below binds to the copy already in the
    # process.  Loaded the other way round, the system runtime and torch's both initialise and this library's
    # sees no device.
    import torch  # noqa: F401
    L = C.CDLL(path)
    L.vamd_create_abi.argtypes = [C.POINTER(_vp), _vp, C.c_size_t, C.c_int, C.c_int]
    L.vamd_quant_limit.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vamd_clock_probe.argtypes = [_vp, _vp]
    L.vamd_config_string.argtypes = [_vp]
    L.vamd_config_string.restype = C.c_char_p
    L.vamd_destroy.argtypes = [_vp]
    L.vamd_destroy.restype = None
    L.vamd_last_error.argtypes = [_vp]
    L.vamd_last_error.restype = C.c_char_p
    L.vamd_set_stream.argtypes = [_vp, _vp]
    L.vamd_input_status.argtypes = [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.vamd_reserve.argtypes = [_vp, C.c_int, C.c_long]
    L.vamd_channels.argtypes = [_vp]
    L.vamd_blocksize.argtypes = [_vp, C.c_int]
    L.vamd_posts.argtypes = [_vp, C.c_int]
    L.vamd_mdct_forward_batch.argtypes = [_vp, C.c_int, _vp, _vp, C.c_long]
    L.vamd_analyze_batch.argtypes = [_vp, C.POINTER(_Desc), C.POINTER(_IO), C.c_int]
    L.vamd_analyze_stream.argtypes = [_vp, C.POINTER(_Desc), C.POINTER(_IO), C.POINTER(C.c_float)]
    L.vamd_analyze_block.argtypes = [_vp, C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp,
                                     _vp, _vp, _vp, _vp, C.POINTER(C.c_float)]
    L.vamd_debug_cycles.argtypes = [_vp, C.c_int, _vp]
    L.vamd_calib_copy.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.vamd_analyze_stream_mixed.argtypes = [_vp, C.POINTER(_Desc), C.POINTER(_IO), C.POINTER(_Desc), C.POINTER(_IO), _vp,
                                            C.c_long, C.POINTER(C.c_float)]
    L.vamd_analyze_streams_mixed.argtypes = [_vp, C.POINTER(_Desc), C.POINTER(_IO), C.POINTER(_Desc), C.POINTER(_IO), _vp, _vp,
                                             C.c_long, C.c_long, _vp]
    L.vamd_envelope_search_batch.argtypes = [_vp, _vp, C.c_long, C.c_long, C.c_long, C.c_long, _vp, _vp]
    L.vamd_envelope_search.argtypes = [_vp, C.POINTER(_vp), C.c_long, _vp, _vp]
    L.vamd_envelope_geometry.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vamd_residue_capacity.argtypes = [_vp, C.c_int]
    L.vamd_analyze_block_res.argtypes = [_vp, C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float] + [_vp] * 10
    L.vamd_analyze_batch_managed.argtypes = [_vp, C.POINTER(_Desc), C.POINTER(_IO), C.POINTER(_MIO)]
    L.vamd_analyze_block_managed.argtypes = [_vp, C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float] + [_vp] * 9
    L.vamd_plan_streams.argtypes = [_vp, _vp, C.c_long, C.c_long, C.c_long, C.c_long, _vp, C.POINTER(_Plan)]
    L.vamd_gather_blocks.argtypes = [_vp, C.POINTER(_Plan), C.c_int, _vp, C.c_long, _vp]
    L.vamd_plan_streams_whole.argtypes = [_vp, _vp, C.c_long, C.c_long, C.c_long, C.c_long, _vp, C.POINTER(_Plan)]
    L.vamd_plan_streams_whole_v.argtypes = [_vp, _vp, C.c_long, C.c_long, C.c_long, C.c_long, _vp, _vp, C.POINTER(_Plan)]
    L.vamd_feed_wrote_v.argtypes = [_vp, C.c_int, C.c_long, _vp]
    L.vamd_feed_create.argtypes = [C.POINTER(_vp), _vp, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_long, C.c_long, C.c_int]
    L.vamd_feed_destroy.argtypes = [_vp]
    L.vamd_feed_destroy.restype = None
    L.vamd_feed_lanes.argtypes = [_vp]
    L.vamd_feed_device.argtypes = [_vp, C.c_int]
    L.vamd_feed_buffer.argtypes = [_vp, C.POINTER(_vp)]
    L.vamd_feed_wrote.argtypes = [_vp, C.c_int, C.c_long, C.c_long]
    L.vamd_feed_packets.argtypes = [_vp, C.c_int, C.POINTER(_FeedResult)]
    L.vamd_feed_release.argtypes = [_vp, C.c_int]
    L.vamd_feed_last_error.argtypes = [_vp]
    L.vamd_feed_last_error.restype = C.c_char_p
    L.vamd_plan_fetch.argtypes = [_vp, C.POINTER(_Plan), _vp * 2, _vp * 2, _vp * 2, _vp * 2, _vp, _vp]
    L.vamd_packet_capacity.argtypes = [_vp, C.c_int]
    L.vamd_submaps.argtypes = [_vp, C.c_int]
    L.vamd_residue_offset.argtypes = [_vp, C.c_int, C.c_int]
    L.vamd_encode_block.argtypes = [_vp, C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _vp, _vp,
                                    C.c_long, _vp]
    L.vamd_encode_blocks.argtypes = [_vp, C.c_long, C.POINTER(_vp), _vp, _vp, _vp, _vp, C.c_float, C.c_int, _vp, _vp, _vp, C.c_long, _vp,
                                     _vp]
    L.vamd_profile.argtypes = [_vp, C.c_int]
    L.vamd_stage_ms.argtypes = [_vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    _lib = L
    return L


def default_setup_blob(name="44k_stereo_q4"):
    """A committed setup blob (vorbis_amd/data/setup_<name>.bin), produced by the reference's own
    libvorbisenc + vorbis_analysis_init through integration/vamd_pack_setup.c (tools/make_setup_blobs.py)."""
    path = os.path.join(_HERE, "data", "setup_%s.bin" % name)
    return np.fromfile(path, dtype=np.uint8)


_FLOAT_OUT = ("mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct")
_INT_OUT = ("ilogmask", "iwork")


class Analyzer:
    """One vamd_ctx: the GPU-side counterpart of a vorbis_dsp_state's lookups."""

    def __init__(self, setup_blob, device=None):
        import torch
        self.torch = torch
        self.L = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("vorbis_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        blob = np.ascontiguousarray(np.frombuffer(bytes(setup_blob), dtype=np.uint8) if not isinstance(setup_blob, np.ndarray)
                                    else setup_blob.astype(np.uint8))
        h = _vp()
        r = self.L.vamd_create_abi(C.byref(h), blob.ctypes.data_as(_vp), blob.size, self.device, ABI_VERSION)
        if r != VAMD_OK:
            raise VamdError(r, "vamd_create failed (see stderr)")
        self.h = h
        self.channels = self.L.vamd_channels(h)
        self.blocksizes = (self.L.vamd_blocksize(h, 0), self.L.vamd_blocksize(h, 1))
        self.posts = (self.L.vamd_posts(h, 0), self.L.vamd_posts(h, 1))

    def close(self):
        if getattr(self, "h", None):
            self.L.vamd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, r):
        if r != VAMD_OK:
            raise VamdError(r, self.L.vamd_last_error(self.h).decode())

    def _dev(self):
        return self.torch.device("cuda", self.device)

    def _need(self, ok, what):
        """Argument checks that survive `python -O`: raw pointers go to the kernels, so a wrong dtype, device or
        stride must stop here."""
        if not ok:
            raise ValueError("vorbis_amd: " + what)

    def _need_tensor(self, v, dtype, name, numel=None, shape=None):
        t = self.torch
        self._need(t.is_tensor(v), "%s must be a torch tensor" % name)
        self._need(v.is_cuda and v.device.index == self.device, "%s must live on cuda:%d (it is on %s)" % (name, self.device, v.device))
        self._need(dtype is None or v.dtype == dtype, "%s must be %s (it is %s)" % (name, dtype, v.dtype))
        self._need(v.is_contiguous(), "%s must be contiguous" % name)
        if numel is not None:
            self._need(v.numel() == numel, "%s must hold %d elements (it holds %d)" % (name, numel, v.numel()))
        if shape is not None:
            self._need(tuple(v.shape) == tuple(shape), "%s must have shape %s (it has %s)" % (name, tuple(shape), tuple(v.shape)))

    def _bind_stream(self):
        s = self.torch.cuda.current_stream(self.device)
        self._check(self.L.vamd_set_stream(self.h, _vp(s.cuda_stream)))

    STAGES = ("transform", "ampmax", "noisemask", "tonemask", "floor", "couple", "residue", "pack")

    def profile(self, enable=True):
        self._bind_stream()
        self._check(self.L.vamd_profile(self.h, 1 if enable else 0))

    def stage_ms(self):
        """(dict stage -> summed ms, number of batches) since the last call; synchronises."""
        ms = (C.c_float * 8)()
        runs = C.c_int(0)
        self._check(self.L.vamd_stage_ms(self.h, ms, 8, C.byref(runs)))
        return dict(zip(self.STAGES, [float(x) for x in ms])), runs.value

    def clock_probe(self, acc):
        """vamd_clock_probe(): while `acc` (a zeroed cuda int64 tensor of 3: shader ticks, 100 MHz ticks, samples) is
        set, every large batch adds a sample of the shader clock taken beside its noise mask; None switches it off.
        After a synchronise: acc[0] / acc[1] * 0.1 = the shader clock in GHz."""
        if acc is not None:
            self._need_tensor(acc, self.torch.int64, "acc", numel=3)
        self._clk = acc   # (kept alive: the library holds the raw pointer)
        self._check(self.L.vamd_clock_probe(self.h, _vp(acc.data_ptr()) if acc is not None else None))

    def debug_cycles(self, enable=True, read=False):
        """Arm/disarm the in-kernel phase stopwatch; with read=True returns the 5x16 tick sums first."""
        out = np.zeros(80, dtype=np.uint64)
        self._check(self.L.vamd_debug_cycles(self.h, 1 if enable else 0, _vp(out.ctypes.data) if read else None))
        return out.reshape(5, 16)

    def calib_copy(self, dst, src):
        """vamd_calib_copy(): src -> dst (device tensors of equal byte size) with the library's named copy kernel."""
        nbytes = src.numel() * src.element_size()
        assert dst.numel() * dst.element_size() == nbytes
        self._check(self.L.vamd_calib_copy(self.h, _vp(dst.data_ptr()), _vp(src.data_ptr()), nbytes))

    def reserve(self, W, max_blocks):
        self._check(self.L.vamd_reserve(self.h, W, max_blocks))

    def input_status(self):
        """vamd_input_status(): synchronise, then (channel-blocks, detector steps) issued since the last call that
        were outside the input domain (a NaN / Inf sample, or quantised values beyond quant_limit()); resets the
        counts.  (0, 0) means every result since then is the reference's, bit for bit.  `last_input_code` keeps the
        call's verdict: VAMD_OK, VAMD_EDOMAIN (finite, beyond the integer bound) or VAMD_ENONFINITE."""
        a, b = C.c_long(0), C.c_long(0)
        r = self.L.vamd_input_status(self.h, C.byref(a), C.byref(b))
        if r not in (VAMD_OK, VAMD_EDOMAIN, VAMD_ENONFINITE):
            self._check(r)
        self.last_input_code = r
        return int(a.value), int(b.value)

    def config_string(self):
        """vamd_config_string(): the environment knobs this context read at vamd_create, as "NAME=value" words."""
        return self.L.vamd_config_string(self.h).decode()

    def quant_limit(self, W, channel=0):
        """vamd_quant_limit(): (Q, first_bin, end_bin, square_bin) -- Q bounds |quantised value| of `channel` at the bins
        [first_bin, end_bin) its residue codes, QUANT_LIMIT_SQUARE those from square_bin on (noise normalisation),
        QUANT_LIMIT_INT all of them: up to there the reference's integer arithmetic is defined."""
        a, b, sq = C.c_int(0), C.c_int(0), C.c_int(0)
        q = int(self.L.vamd_quant_limit(self.h, W, channel, C.byref(a), C.byref(b), C.byref(sq)))
        self._need(q >= 0, "vamd_quant_limit(%d, %d)" % (W, channel))
        return q, a.value, b.value, sq.value

    def beyond_quant_limit(self, W, iwork, residue=True):
        """Which channels of a block's quantised values iwork[ch][n/2] (a numpy array; e.g. the reference's own) lie
        outside the input domain's integer edge: uint8 [ch] of STATUS_RANGE / 0 -- what `status` must say of them.
        residue=False: the call ran no residue search (no res_* / packets asked for), so only the coupling stage's
        bounds were tested."""
        out = np.zeros(iwork.shape[0], np.uint8)
        for c in range(iwork.shape[0]):
            q, lo, hi, sq = self.quant_limit(W, c)
            v = np.abs(iwork[c].astype(np.int64))
            lim = np.full(v.shape, QUANT_LIMIT_INT, np.int64)
            lim[sq:] = QUANT_LIMIT_SQUARE
            if residue:
                lim[lo:hi] = np.minimum(lim[lo:hi], q)
            out[c] = STATUS_RANGE if (v > lim).any() else 0
        return out

    # mdct_forward(lookup, in, out) batched -- BASELINE config 2
    def mdct_forward(self, W, frames, out=None):
        t = self.torch
        n = self.blocksizes[W]
        self._need_tensor(frames, t.float32, "frames")
        self._need(frames.dim() >= 1 and frames.shape[-1] == n, "frames must end in %d samples" % n)
        nf = frames.numel() // n
        if out is None:
            out = t.empty(frames.shape[:-1] + (n // 2,), dtype=t.float32, device=frames.device)
        else:
            self._need_tensor(out, t.float32, "out", numel=nf * (n // 2))
        self._bind_stream()
        self._check(self.L.vamd_mdct_forward_batch(self.h, W, _vp(frames.data_ptr()), _vp(out.data_ptr()), nf))
        return out

    def _desc(self, W, nb, lW, nW, blocktype, ampmax_in, keep):
        t = self.torch
        d = _Desc()
        d.W, d.nblocks = W, nb

        def arr(v, dtype, name, uni):
            if isinstance(v, (list, tuple, np.ndarray)):   # per-block values from the host
                v = t.as_tensor(np.asarray(v), dtype=dtype).to(self._dev())
            if t.is_tensor(v):
                self._need_tensor(v, dtype, name, numel=nb)
                keep.append(v)
                setattr(d, name, _vp(v.data_ptr()))
            else:
                setattr(d, name, None)
                setattr(d, uni, v)
        arr(lW, t.int32, "lW", "uniform_lW")
        arr(nW, t.int32, "nW", "uniform_nW")
        arr(blocktype, t.int32, "blocktype", "uniform_blocktype")
        arr(ampmax_in, t.float32, "ampmax_in", "uniform_ampmax_in")
        return d

    def alloc_outputs(self, W, nb, want):
        """Output tensors for `want` (names from vamd_batch_io)."""
        t = self.torch
        ch, n2 = self.channels, self.blocksizes[W] // 2
        dev = self._dev()
        o = {}
        for k in want:
            if k in _FLOAT_OUT:
                o[k] = t.empty((nb, ch, n2), dtype=t.float32, device=dev)
            elif k in _INT_OUT:
                o[k] = t.empty((nb, ch, n2), dtype=t.int32, device=dev)
            elif k == "posts":
                o[k] = t.empty((nb, ch, POSTS_STRIDE), dtype=t.int32, device=dev)
            elif k in ("post_valid", "nonzero"):
                o[k] = t.empty((nb, ch), dtype=t.int32, device=dev)
            elif k == "local_ampmax":
                o[k] = t.empty((nb, ch), dtype=t.float32, device=dev)
            elif k == "ampmax_out":
                o[k] = t.empty((nb,), dtype=t.float32, device=dev)
            elif k == "res_class":   # (modes with two submaps: one more axis, [nb, 2, ...], here and in res_count)
                o[k] = t.zeros((nb,) + self._sub(W) + (RES_CLASS_STRIDE,), dtype=t.int32, device=dev)
            elif k == "res_entries":
                o[k] = t.zeros((nb, self.residue_capacity(W)), dtype=t.int16, device=dev)  # uint16 payload
            elif k == "res_count":
                o[k] = t.zeros((nb,) + self._sub(W) + (2,), dtype=t.int32, device=dev)
            elif k == "packets":
                o[k] = t.zeros((nb, self.packet_capacity(W)), dtype=t.uint8, device=dev)
            elif k == "packet_bits":
                o[k] = t.zeros((nb,), dtype=t.int32, device=dev)
            elif k == "status":      # STATUS_* bits: the channel-block was outside the input domain (vorbis_amd.h)
                o[k] = t.zeros((nb, ch), dtype=t.uint8, device=dev)
            else:
                raise KeyError(k)
        return o

    def submaps(self, W):
        return int(self.L.vamd_submaps(self.h, W))

    def _sub(self, W):
        S = self.submaps(W)
        return (S,) if S > 1 else ()

    def residue_lists(self, W, cls, ent, cnt):
        """One block's residue rows -> (classes, entries) with the submaps one after the other, the order the
        reference classifies / emits them in.  cls / ent / cnt: numpy rows of res_class / res_entries / res_count."""
        S = self.submaps(W)
        cls, cnt = np.asarray(cls).reshape(S, RES_CLASS_STRIDE), np.asarray(cnt).reshape(S, 2)
        ent = np.asarray(ent).view(np.uint16)
        offs = [int(self.L.vamd_residue_offset(self.h, W, sm)) for sm in range(S)]
        return (np.concatenate([cls[sm, :cnt[sm, 0]] for sm in range(S)]),
                np.concatenate([ent[offs[sm]:offs[sm] + cnt[sm, 1]] for sm in range(S)]))

    def residue_capacity(self, W):
        """Row length of res_entries for size class W; 0 when the GPU does not cover this mode's residue."""
        return int(self.L.vamd_residue_capacity(self.h, W))

    def packet_capacity(self, W):
        """Bytes the longest possible packet of size class W takes; 0 when packets are not assembled on the GPU."""
        return int(self.L.vamd_packet_capacity(self.h, W))

    _OUT_DTYPES = None

    def _out_dtype(self, k):
        t = self.torch
        if k in _FLOAT_OUT or k in ("local_ampmax", "ampmax_out"):
            return t.float32
        if k == "res_entries":
            return t.int16
        if k in ("packets", "status"):
            return t.uint8
        return t.int32

    def _io(self, pcm, outs, nb=None):
        io = _IO()
        io.pcm = _vp(pcm.data_ptr())
        nb = pcm.shape[0] if nb is None else nb
        for k, v in outs.items():
            self._need(hasattr(io, k), "unknown output %r" % (k,))
            self._need_tensor(v, self._out_dtype(k), "outs[%r]" % k)
            self._need(v.dim() >= 1 and v.shape[0] == nb, "outs[%r] must have one row per block (%d)" % (k, nb))
            setattr(io, k, _vp(v.data_ptr()))
        if "packets" in outs:
            io.packet_stride = outs["packets"].shape[-1]
        return io

    _DEFAULT_WANT = {LEVEL_TRANSFORM: ("mdct_raw", "logfft", "logmdct", "local_ampmax"),
                     LEVEL_PSY: ("mdct_raw", "noise", "tone"),
                     LEVEL_FULL: ("mdct", "logmask", "posts", "post_valid", "iwork", "nonzero", "ampmax_out")}

    def analyze(self, pcm, W=1, lW=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0, level=LEVEL_FULL,
                want=None, outs=None):
        """vamd_analyze_batch.  pcm: cuda float32 [nblocks, ch, n].  Returns dict name -> tensor."""
        t = self.torch
        n = self.blocksizes[W]
        self._need(W in (0, 1), "W must be 0 or 1")
        self._need_tensor(pcm, t.float32, "pcm")
        self._need(pcm.dim() == 3 and pcm.shape[1] == self.channels and pcm.shape[2] == n,
                   "pcm must be [blocks, %d, %d] (it is %s)" % (self.channels, n, tuple(pcm.shape)))
        nb = pcm.shape[0]
        if outs is None:
            outs = self.alloc_outputs(W, nb, self._DEFAULT_WANT[level] if want is None else want)
        keep = []
        d = self._desc(W, nb, lW, nW, blocktype, ampmax_in, keep)
        io = self._io(pcm, outs)
        self._bind_stream()
        self._check(self.L.vamd_analyze_batch(self.h, C.byref(d), C.byref(io), level))
        return outs

    def analyze_stream(self, pcm, ampmax_state, W=1, lW=1, nW=1, blocktype=BLOCKTYPE_LONG, want=None, outs=None):
        """vamd_analyze_stream.  Returns (outs, new ampmax_state)."""
        n = self.blocksizes[W]
        self._need_tensor(pcm, self.torch.float32, "pcm")
        self._need(pcm.dim() == 3 and tuple(pcm.shape[1:]) == (self.channels, n),
                   "pcm must be [blocks, %d, %d] (it is %s)" % (self.channels, n, tuple(pcm.shape)))
        nb = pcm.shape[0]
        if outs is None:
            outs = self.alloc_outputs(W, nb, self._DEFAULT_WANT[LEVEL_FULL] if want is None else want)
        keep = []
        d = self._desc(W, nb, lW, nW, blocktype, 0.0, keep)
        io = self._io(pcm, outs)
        st = C.c_float(ampmax_state)
        self._bind_stream()
        self._check(self.L.vamd_analyze_stream(self.h, C.byref(d), C.byref(io), C.byref(st)))
        return outs, st.value

    def analyze_stream_mixed(self, blocks, ampmax_state, want=None):
        """vamd_analyze_stream_mixed.  `blocks`: stream-ordered list of dicts with keys pcm (numpy
        [ch][n]), W, lW, nW, blocktype.  Returns (per-block list of output dicts in stream order, state)."""
        t = self.torch
        want = self._DEFAULT_WANT[LEVEL_FULL] if want is None else want
        idx = {0: [], 1: []}
        order = []
        for b in blocks:
            order.append((b["W"] << 30) | len(idx[b["W"]]))
            idx[b["W"]].append(b)
        descs, ios, outs, keep = {}, {}, {}, []
        for W in (0, 1):
            sel = idx[W]
            nb = len(sel)
            n = self.blocksizes[W]
            pcm = t.from_numpy(np.stack([b["pcm"] for b in sel]).astype(np.float32)).to(self._dev()) if nb else \
                t.empty((0, self.channels, n), device=self._dev())
            dv = lambda k: t.tensor([b[k] for b in sel], dtype=t.int32, device=self._dev()) if nb else 0  # noqa: E731
            outs[W] = self.alloc_outputs(W, nb, want)
            descs[W] = self._desc(W, nb, dv("lW"), dv("nW"), dv("blocktype"), 0.0, keep)
            ios[W] = self._io(pcm, outs[W])
            keep.append(pcm)
        od = t.tensor(order, dtype=t.int32, device=self._dev())
        st = C.c_float(ampmax_state)
        self._bind_stream()
        self._check(self.L.vamd_analyze_stream_mixed(self.h, C.byref(descs[0]), C.byref(ios[0]), C.byref(descs[1]),
                                                     C.byref(ios[1]), _vp(od.data_ptr()), len(order), C.byref(st)))
        host = {W: {k: v.cpu().numpy() for k, v in outs[W].items()} for W in (0, 1)}
        res = []
        for o in order:
            W, i = (o >> 30) & 1, o & 0x3fffffff
            res.append({k: v[i] for k, v in host[W].items()})
        return res, st.value

    def analyze_streams_mixed(self, streams, ampmax_states, want=None):
        """vamd_analyze_streams_mixed.  `streams`: list of stream-ordered block lists (dicts as for
        analyze_stream_mixed); `ampmax_states`: one float per stream.  One call for all of them.
        Returns (per stream: per-block list of output dicts, new states as numpy)."""
        t = self.torch
        want = self._DEFAULT_WANT[LEVEL_FULL] if want is None else want
        idx = {0: [], 1: []}
        order, start = [], [0]
        for blocks in streams:
            for b in blocks:
                order.append((b["W"] << 30) | len(idx[b["W"]]))
                idx[b["W"]].append(b)
            start.append(len(order))
        descs, ios, outs, keep = {}, {}, {}, []
        for W in (0, 1):
            sel = idx[W]
            nb = len(sel)
            n = self.blocksizes[W]
            pcm = t.from_numpy(np.stack([b["pcm"] for b in sel]).astype(np.float32)).to(self._dev()) if nb else \
                t.empty((0, self.channels, n), device=self._dev())
            dv = lambda k: t.tensor([b[k] for b in sel], dtype=t.int32, device=self._dev()) if nb else 0  # noqa: E731
            outs[W] = self.alloc_outputs(W, nb, want)
            descs[W] = self._desc(W, nb, dv("lW"), dv("nW"), dv("blocktype"), 0.0, keep)
            ios[W] = self._io(pcm, outs[W])
            keep.append(pcm)
        od = t.tensor(order, dtype=t.int32, device=self._dev())
        sd = t.tensor(start, dtype=t.int64, device=self._dev())
        st = t.tensor(list(ampmax_states), dtype=t.float32, device=self._dev())
        self._bind_stream()
        self._check(self.L.vamd_analyze_streams_mixed(self.h, C.byref(descs[0]), C.byref(ios[0]), C.byref(descs[1]),
                                                      C.byref(ios[1]), _vp(od.data_ptr()), _vp(sd.data_ptr()), len(streams),
                                                      len(order), _vp(st.data_ptr())))
        t.cuda.synchronize()
        host = {W: {k: v.cpu().numpy() for k, v in outs[W].items()} for W in (0, 1)}
        res, k = [], 0
        for blocks in streams:
            cur = []
            for _ in blocks:
                o = order[k]
                W, i = (o >> 30) & 1, o & 0x3fffffff
                cur.append({kk: v[i] for kk, v in host[W].items()})
                k += 1
            res.append(cur)
        return res, st.cpu().numpy()

    # ---- device-resident stream control: whole streams in, block lists out (vamd_plan_streams) ----
    def plan_streams(self, streams, states=None):
        """streams: cuda float32 [nstreams, ch, nsamples], each laid out as the encoder's own PCM buffer (first block
        centred at blocksizes[1]/2).  Returns (plan, states): the plan struct (its device arrays belong to the
        context and stay valid until the next plan_streams) and the detector states tensor."""
        t = self.torch
        self._need_tensor(streams, t.float32, "streams")
        self._need(streams.dim() == 3 and streams.shape[1] == self.channels, "streams must be [nstreams, %d, nsamples]" % self.channels)
        ns, ch, ln = streams.shape
        if states is None:
            states = t.zeros((ns, C.sizeof(EnvelopeState)), dtype=t.uint8, device=self._dev())
        plan = _Plan()
        self._bind_stream()
        self._check(self.L.vamd_plan_streams(self.h, _vp(streams.data_ptr()), ch * ln, ln, ns, ln, _vp(states.data_ptr()),
                                             C.byref(plan)))
        return plan, states

    def plan_streams_whole(self, streams, nframes, states=None):
        """vamd_plan_streams_whole: COMPLETE streams.  streams: cuda float32 [nstreams, ch, row], every channel row laid
        out [blocksizes[1]/2 of room | nframes real samples | >= 3 * blocksizes[1] of room]; the call fills the room
        either end with the reference's LPC extrapolations (lib/block.c:417-458, :474-512), so `streams` is written.
        Returns (plan, states) as plan_streams; the plan runs to each stream's last block."""
        t = self.torch
        self._need_tensor(streams, t.float32, "streams")
        self._need(streams.dim() == 3 and streams.shape[1] == self.channels, "streams must be [nstreams, %d, row]" % self.channels)
        ns, ch, ln = streams.shape
        self._need(ln % 4 == 0 and ln >= self.blocksizes[1] // 2 + int(np.max(nframes)) + 3 * self.blocksizes[1],
                   "a channel row needs blocksizes[1]/2 + nframes + 3 * blocksizes[1] samples (a multiple of 4)")
        if states is None:
            states = t.zeros((ns, C.sizeof(EnvelopeState)), dtype=t.uint8, device=self._dev())
        plan = _Plan()
        self._bind_stream()
        if np.ndim(nframes) == 0:
            self._check(self.L.vamd_plan_streams_whole(self.h, _vp(streams.data_ptr()), ch * ln, ln, ns, int(nframes), _vp(states.data_ptr()),
                                                       C.byref(plan)))
        else:  # streams of unequal length: every buffer laid out for the longest, zero behind a shorter stream's samples
            fr = np.ascontiguousarray(nframes, dtype=np.int64)
            self._need(fr.shape == (ns,), "nframes must hold one length per stream")
            self._check(self.L.vamd_plan_streams_whole_v(self.h, _vp(streams.data_ptr()), ch * ln, ln, ns, int(fr.max()), _vp(fr.ctypes.data),
                                                         _vp(states.data_ptr()), C.byref(plan)))
        return plan, states

    def plan_lists(self, plan):
        """A plan on the host: dict with per size class lW / nW / blocktype / src arrays, order and stream_start."""
        out = {}
        arrs = {}
        for name, dt in (("lW", np.int32), ("nW", np.int32), ("blocktype", np.int32), ("src", np.int64)):
            arrs[name] = [np.zeros(max(1, plan.nblocks[W]), dt) for W in (0, 1)]
        order = np.zeros(max(1, plan.nblocks[0] + plan.nblocks[1]), np.int32)
        start = np.zeros(plan.nstreams + 1, np.int64)
        pair = lambda k: (_vp * 2)(_vp(arrs[k][0].ctypes.data), _vp(arrs[k][1].ctypes.data))  # noqa: E731
        self._check(self.L.vamd_plan_fetch(self.h, C.byref(plan), pair("lW"), pair("nW"), pair("blocktype"), pair("src"),
                                           _vp(order.ctypes.data), _vp(start.ctypes.data)))
        for k, v in arrs.items():
            out[k] = [v[W][:plan.nblocks[W]] for W in (0, 1)]
        out["order"] = order[:plan.nblocks[0] + plan.nblocks[1]]
        out["stream_start"] = start
        return out

    def gather_blocks(self, plan, W, streams, out=None):
        """The planned blocks of size class W copied out of `streams` into [nblocks[W], ch, blocksize[W]]."""
        t = self.torch
        n = self.blocksizes[W]
        # the layout the plan was made for: float32, on this device, contiguous [nstreams, ch, nsamples], 16-byte rows
        self._need_tensor(streams, t.float32, "streams")
        self._need(streams.dim() == 3 and streams.shape[1] == self.channels and streams.shape[0] == plan.nstreams,
                   "streams must be the [%d, %d, nsamples] tensor the plan was made from" % (plan.nstreams, self.channels))
        self._need(streams.shape[2] % 4 == 0, "nsamples must be a multiple of 4 (16-byte lanes)")
        if out is None:
            out = t.empty((plan.nblocks[W], self.channels, n), dtype=t.float32, device=self._dev())
        else:
            self._need_tensor(out, t.float32, "out", numel=plan.nblocks[W] * self.channels * n)
        self._bind_stream()
        self._check(self.L.vamd_gather_blocks(self.h, C.byref(plan), W, _vp(streams.data_ptr()), streams.shape[2], _vp(out.data_ptr())))
        return out

    def analyze_plan(self, plan, pcm_blocks, outs, ampmax_states, streams=None):
        """vamd_analyze_streams_mixed over a plan: outs = per size class (index 0 short, 1 long) the output dict;
        ampmax_states: cuda float32 [nstreams], updated in place.  The blocks' samples: either pcm_blocks = per size class
        the gathered batch (gather_blocks), or -- pcm_blocks None -- `streams`, the very [nstreams, ch, nsamples] tensor the
        plan was made from, read in place through the plan's offsets (vamd_batch_io::pcm_src: no gathered copy)."""
        t = self.torch
        descs, ios = [], []
        if pcm_blocks is None:
            self._need_tensor(streams, t.float32, "streams")
            self._need(streams.dim() == 3 and streams.shape[1] == self.channels and streams.shape[0] == plan.nstreams and
                       streams.shape[2] % 4 == 0, "streams must be the [%d, %d, nsamples] tensor the plan was made from" % (plan.nstreams, self.channels))
        for W in (0, 1):
            d = _Desc()
            d.W, d.nblocks = W, plan.nblocks[W]
            d.lW, d.nW, d.blocktype, d.ampmax_in = plan.lW[W], plan.nW[W], plan.blocktype[W], None
            descs.append(d)
            if not plan.nblocks[W]:
                ios.append(_IO())
            elif pcm_blocks is None:
                io = self._io(streams, outs[W], nb=plan.nblocks[W])
                io.pcm_src, io.pcm_channel_stride = plan.src[W], streams.shape[2]
                ios.append(io)
            else:
                self._need_tensor(pcm_blocks[W], t.float32, "pcm_blocks[%d]" % W, numel=plan.nblocks[W] * self.channels * self.blocksizes[W])
                ios.append(self._io(pcm_blocks[W].reshape(plan.nblocks[W], self.channels, self.blocksizes[W]), outs[W]))
        self._need_tensor(ampmax_states, t.float32, "ampmax_states", numel=plan.nstreams)
        self._bind_stream()
        self._check(self.L.vamd_analyze_streams_mixed(self.h, C.byref(descs[0]), C.byref(ios[0]), C.byref(descs[1]), C.byref(ios[1]),
                                                      plan.order, plan.stream_start, plan.nstreams, plan.nblocks[0] + plan.nblocks[1],
                                                      _vp(ampmax_states.data_ptr())))

    def analyze_block(self, pcm, lW=1, W=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0):
        """vamd_analyze_block: host numpy pcm[ch][n] in, host numpy results out (the per-block
        compatibility path that sits behind vorbis_analysis())."""
        ch, n = self.channels, self.blocksizes[W]
        n2 = n // 2
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        self._need(pcm.shape == (ch, n), "pcm must be [%d][%d] (it is %s)" % (ch, n, pcm.shape))
        ptrs = (_vp * ch)(*[_vp(pcm[i].ctypes.data) for i in range(ch)])
        o = dict(mdct=np.empty((ch, n2), np.float32), logmask=np.empty((ch, n2), np.float32),
                 posts=np.empty((ch, POSTS_STRIDE), np.int32), post_valid=np.empty(ch, np.int32),
                 iwork=np.empty((ch, n2), np.int32), nonzero=np.empty(ch, np.int32))
        amp = C.c_float(0)
        self._bind_stream()
        cap = self.residue_capacity(W)
        if cap > 0:  # the residue back-end's decisions ride along where the mode is covered
            S = self.submaps(W)
            rcls, rent, rcnt = np.zeros(S * RES_CLASS_STRIDE, np.int32), np.zeros(cap, np.uint16), np.zeros(2 * S, np.int32)
            self._check(self.L.vamd_analyze_block_res(
                self.h, ptrs, lW, W, nW, blocktype, ampmax_in, _vp(o["mdct"].ctypes.data), _vp(o["logmask"].ctypes.data),
                _vp(o["posts"].ctypes.data), _vp(o["post_valid"].ctypes.data), _vp(o["iwork"].ctypes.data),
                _vp(o["nonzero"].ctypes.data), C.cast(C.byref(amp), _vp), _vp(rcls.ctypes.data), _vp(rent.ctypes.data),
                _vp(rcnt.ctypes.data)))
            o["res_class"], o["res_entries"] = self.residue_lists(W, rcls, rent, rcnt)
        else:
            self._check(self.L.vamd_analyze_block(self.h, ptrs, lW, W, nW, blocktype, ampmax_in,
                                                  _vp(o["mdct"].ctypes.data), _vp(o["logmask"].ctypes.data),
                                                  _vp(o["posts"].ctypes.data), _vp(o["post_valid"].ctypes.data),
                                                  _vp(o["iwork"].ctypes.data), _vp(o["nonzero"].ctypes.data),
                                                  C.byref(amp)))
        o["ampmax_out"] = amp.value
        return o

    # ---- bitrate-managed blocks: fifteen candidate packets each (vamd_analyze_*_managed) ------------
    def analyze_managed(self, pcm, W=1, lW=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0, residue=False,
                        packets=False, want=()):
        """vamd_analyze_batch_managed.  pcm: cuda float32 [nblocks, ch, n].  Returns a dict: shared `mdct`,
        `logmask`, `ampmax_out`, and per candidate `m_posts` [nb,15,ch,32], `m_post_valid` / `m_nonzero`
        [nb,15,ch], `m_iwork` [nb,15,ch,n/2] (+ `m_res_class`, `m_res_entries`, `m_res_count` with residue=True)."""
        t = self.torch
        n = self.blocksizes[W]
        self._need_tensor(pcm, t.float32, "pcm")
        self._need(pcm.dim() == 3 and tuple(pcm.shape[1:]) == (self.channels, n), "pcm must be [blocks, %d, %d]" % (self.channels, n))
        nb, ch, n2, dev = pcm.shape[0], self.channels, n // 2, pcm.device
        outs = self.alloc_outputs(W, nb, ("mdct", "logmask", "ampmax_out") + tuple(want))  # (want: further shared outputs, e.g. "status")
        keep = []
        d = self._desc(W, nb, lW, nW, blocktype, ampmax_in, keep)
        io = self._io(pcm, outs)
        mo = {"posts": t.empty((nb, PACKETBLOBS, ch, POSTS_STRIDE), dtype=t.int32, device=dev),
              "post_valid": t.empty((nb, PACKETBLOBS, ch), dtype=t.int32, device=dev),
              "iwork": t.empty((nb, PACKETBLOBS, ch, n2), dtype=t.int32, device=dev),
              "nonzero": t.empty((nb, PACKETBLOBS, ch), dtype=t.int32, device=dev)}
        if residue:
            mo["res_class"] = t.zeros((nb, PACKETBLOBS) + self._sub(W) + (RES_CLASS_STRIDE,), dtype=t.int32, device=dev)
            mo["res_entries"] = t.zeros((nb, PACKETBLOBS, self.residue_capacity(W)), dtype=t.int16, device=dev)
            mo["res_count"] = t.zeros((nb, PACKETBLOBS) + self._sub(W) + (2,), dtype=t.int32, device=dev)
        if packets:  # (+ `m_packets` [nb,15,packet_capacity] uint8, `m_packet_bits` [nb,15])
            mo["packets"] = t.zeros((nb, PACKETBLOBS, self.packet_capacity(W)), dtype=t.uint8, device=dev)
            mo["packet_bits"] = t.zeros((nb, PACKETBLOBS), dtype=t.int32, device=dev)
        m = _MIO()
        for k, v in mo.items():
            setattr(m, k, _vp(v.data_ptr()))
        if packets:
            m.packet_stride = mo["packets"].shape[-1]
        self._bind_stream()
        self._check(self.L.vamd_analyze_batch_managed(self.h, C.byref(d), C.byref(io), C.byref(m)))
        for k, v in mo.items():
            outs["m_" + k] = v
        return outs

    def analyze_block_managed(self, pcm, lW=1, W=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0):
        """vamd_analyze_block_managed: host numpy pcm[ch][n] in; shared mdct / ampmax_out and the fifteen
        candidates' m_posts / m_post_valid / m_iwork / m_nonzero (+ residue decisions where covered) out."""
        ch, n = self.channels, self.blocksizes[W]
        n2 = n // 2
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        self._need(pcm.shape == (ch, n), "pcm must be [%d][%d] (it is %s)" % (ch, n, pcm.shape))
        ptrs = (_vp * ch)(*[_vp(pcm[i].ctypes.data) for i in range(ch)])
        o = dict(mdct=np.empty((ch, n2), np.float32), m_posts=np.empty((PACKETBLOBS, ch, POSTS_STRIDE), np.int32),
                 m_post_valid=np.empty((PACKETBLOBS, ch), np.int32), m_iwork=np.empty((PACKETBLOBS, ch, n2), np.int32),
                 m_nonzero=np.empty((PACKETBLOBS, ch), np.int32))
        amp = C.c_float(0)
        cap = self.residue_capacity(W)
        S = self.submaps(W)
        rcls = np.zeros((PACKETBLOBS, S * RES_CLASS_STRIDE), np.int32)
        rent = np.zeros((PACKETBLOBS, max(cap, 1)), np.uint16)
        rcnt = np.zeros((PACKETBLOBS, 2 * S), np.int32)
        rp = [_vp(rcls.ctypes.data), _vp(rent.ctypes.data), _vp(rcnt.ctypes.data)] if cap > 0 else [None, None, None]
        self._bind_stream()
        self._check(self.L.vamd_analyze_block_managed(
            self.h, ptrs, lW, W, nW, blocktype, ampmax_in, _vp(o["mdct"].ctypes.data), C.cast(C.byref(amp), _vp),
            _vp(o["m_posts"].ctypes.data), _vp(o["m_post_valid"].ctypes.data), _vp(o["m_iwork"].ctypes.data),
            _vp(o["m_nonzero"].ctypes.data), *rp))
        o["ampmax_out"] = amp.value
        if cap > 0:
            lists = [self.residue_lists(W, rcls[k], rent[k], rcnt[k]) for k in range(PACKETBLOBS)]
            o["m_res_class"] = [l[0] for l in lists]
            o["m_res_entries"] = [l[1] for l in lists]
        return o

    def encode_block(self, pcm, lW=1, W=1, nW=1, blocktype=BLOCKTYPE_LONG, ampmax_in=-9999.0, managed=False):
        """vamd_encode_block: host numpy pcm[ch][n] in; (list of 1 or 15 packets as bytes, ampmax_out) out."""
        ch, n = self.channels, self.blocksizes[W]
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        self._need(pcm.shape == (ch, n), "pcm must be [%d][%d] (it is %s)" % (ch, n, pcm.shape))
        ptrs = (_vp * ch)(*[_vp(pcm[i].ctypes.data) for i in range(ch)])
        nk, cap = (PACKETBLOBS if managed else 1), self.packet_capacity(W)
        pk, bits = np.zeros((nk, max(cap, 4)), np.uint8), np.zeros(nk, np.int32)
        amp = C.c_float(0)
        self._bind_stream()
        self._check(self.L.vamd_encode_block(self.h, ptrs, lW, W, nW, blocktype, C.c_float(ampmax_in), 1 if managed else 0,
                                             C.cast(C.byref(amp), _vp), _vp(pk.ctypes.data), C.c_long(pk.shape[1]),
                                             _vp(bits.ctypes.data)))
        return [packet_bytes(pk[k], bits[k]) for k in range(nk)], amp.value

    def encode_blocks(self, pcm, lW, W, nW, blocktype, ampmax_in_first=-9999.0, managed=False):
        """vamd_encode_blocks: consecutive blocks of ONE stream in one launch sequence.  pcm: list of host numpy
        [ch][blocksize[W[b]]]; lW / W / nW / blocktype: per block.  Returns (packets[b][k] as bytes -- None where the
        block's verdict is not VAMD_OK --, ampmax_in float32[nb], ampmax_out float32[nb], verdict int32[nb])."""
        ch, nb = self.channels, len(pcm)
        self._need(nb == len(lW) == len(W) == len(nW) == len(blocktype), "one lW / W / nW / blocktype per block")
        keep = [np.ascontiguousarray(x, dtype=np.float32) for x in pcm]
        for b in range(nb):
            self._need(int(W[b]) in (0, 1) and keep[b].shape == (ch, self.blocksizes[int(W[b])]),
                       "pcm[%d] must be [%d][blocksize[W]]" % (b, ch))
        ptrs = (_vp * max(1, nb * ch))(*[_vp(keep[b][c].ctypes.data) for b in range(nb) for c in range(ch)])
        arr = [np.ascontiguousarray(v, dtype=np.int32) for v in (lW, W, nW, blocktype)]
        nk, cap = (PACKETBLOBS if managed else 1), max(self.packet_capacity(0), self.packet_capacity(1), 4)
        pk, bits = np.zeros((max(nb, 1), nk, cap), np.uint8), np.zeros((max(nb, 1), nk), np.int32)
        ain, aout, verdict = np.zeros(max(nb, 1), np.float32), np.zeros(max(nb, 1), np.float32), np.zeros(max(nb, 1), np.int32)
        self._bind_stream()
        self._check(self.L.vamd_encode_blocks(self.h, C.c_long(nb), ptrs, *[_vp(a.ctypes.data) for a in arr],
                                              C.c_float(ampmax_in_first), 1 if managed else 0, _vp(ain.ctypes.data),
                                              _vp(aout.ctypes.data), _vp(pk.ctypes.data), C.c_long(cap),
                                              _vp(bits.ctypes.data), _vp(verdict.ctypes.data)))
        packets = [[packet_bytes(pk[b, k], bits[b, k]) for k in range(nk)] if verdict[b] == 0 else None for b in range(nb)]
        return packets, ain[:nb], aout[:nb], verdict[:nb]

    # ---- the block-switching detector (vamd_envelope_search*) ---------------------------------
    def envelope_geometry(self):
        w, s = C.c_int(0), C.c_int(0)
        self._check(self.L.vamd_envelope_geometry(self.h, C.byref(w), C.byref(s)))
        return w.value, s.value

    def envelope_search(self, pcm, nsteps, state=None):
        """vamd_envelope_search: host pcm[ch][>= (nsteps-1)*searchstep + winlength] -> (ret flags uint8[nsteps],
        state).  `state` is an EnvelopeState (None = start of stream) and is updated in place."""
        ch = self.channels
        win, step = self.envelope_geometry()
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        self._need(pcm.ndim == 2 and pcm.shape[0] == ch and pcm.shape[1] >= (nsteps - 1) * step + win,
                   "pcm must be [%d][>= %d samples]" % (ch, (nsteps - 1) * step + win))
        if state is None:
            state = EnvelopeState()
        ret = np.zeros(nsteps, np.uint8)
        ptrs = (_vp * ch)(*[_vp(pcm[i].ctypes.data) for i in range(ch)])
        self._bind_stream()
        self._check(self.L.vamd_envelope_search(self.h, ptrs, nsteps, C.byref(state), _vp(ret.ctypes.data)))
        return ret, state

    def envelope_search_batch(self, pcm, nsteps, states=None, ret=None):
        """vamd_envelope_search_batch: cuda float32 pcm[nstreams][ch][len]; states = cuda uint8 tensor
        [nstreams][sizeof(vamd_envelope_state)] (zeros = fresh streams), updated in place.
        Returns (ret cuda uint8 [nstreams][nsteps], states)."""
        t = self.torch
        self._need_tensor(pcm, t.float32, "pcm")
        self._need(pcm.dim() == 3, "pcm must be [streams, channels, samples]")
        ns, ch, ln = pcm.shape
        self._need(ch == self.channels, "pcm must have %d channels" % self.channels)
        win, step = self.envelope_geometry()
        self._need(ln >= (nsteps - 1) * step + win, "streams too short for %d steps" % nsteps)
        if states is None:
            states = t.zeros((ns, C.sizeof(EnvelopeState)), dtype=t.uint8, device=pcm.device)
        if ret is None:
            ret = t.empty((ns, nsteps), dtype=t.uint8, device=pcm.device)
        self._bind_stream()
        self._check(self.L.vamd_envelope_search_batch(self.h, _vp(pcm.data_ptr()), ch * ln, ln, ns, nsteps,
                                                      _vp(states.data_ptr()), _vp(ret.data_ptr())))
        return ret, states


class EnvelopeState(C.Structure):
    """vamd_envelope_state (include/vorbis_amd.h); all-zero = start of a stream."""
    _fields_ = [("steps", C.c_int64), ("stretch", C.c_int32), ("pad", C.c_int32),
                ("near_hist", C.c_float * 30 * MAX_CH), ("amp_hist", C.c_float * 8 * 16 * MAX_CH)]


FEED_S16, FEED_F32 = 0, 1


class Feed:
    """vamd_feed: whole streams from HOST memory in (interleaved int16 / float32), finished packets back to host
    memory, over one or several GPUs (include/vorbis_amd.h, "the host-fed farm").  The call sequence is libvorbis'
    own, for a group of streams: buffer() -> fill -> wrote() -> packets() -> release()."""

    def __init__(self, setup_blob, devices=None, lanes_per_device=2, max_streams=256, max_frames=131072, fmt=FEED_S16):
        self.L = load_library()
        blob = np.ascontiguousarray(setup_blob, dtype=np.uint8)
        devs = list(devices) if devices else []
        arr = (C.c_int * max(1, len(devs)))(*devs)
        h = _vp()
        r = self.L.vamd_feed_create(C.byref(h), _vp(blob.ctypes.data), blob.size, arr if devs else None, len(devs), lanes_per_device,
                                    max_streams, max_frames, fmt)
        if r:
            raise VamdError(r, "vamd_feed_create failed (setup without GPU-assembled packets, bad arguments, or a HIP failure)")
        self.h = h
        self.max_streams, self.max_frames, self.fmt = max_streams, max_frames, fmt
        self.dtype = np.int16 if fmt == FEED_S16 else np.float32
        self.lanes = self.L.vamd_feed_lanes(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.vamd_feed_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, r):
        if r < 0:
            raise VamdError(r, self.L.vamd_feed_last_error(self.h).decode())
        return r

    def device(self, slot):
        return self._check(self.L.vamd_feed_device(self.h, slot))

    def buffer(self, channels):
        """-> (slot, array [max_streams * max_frames * channels] of the feed's sample type over the lane's pinned input arena)"""
        p = _vp()
        slot = self._check(self.L.vamd_feed_buffer(self.h, C.byref(p)))
        n = self.max_streams * self.max_frames * channels
        ct = C.c_int16 if self.fmt == FEED_S16 else C.c_float
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,))
        return slot, arr

    def wrote(self, slot, nstreams, frames):
        """frames: one length for every stream, or one per stream (the streams then lie back to back in the arena)"""
        if np.ndim(frames) == 0:
            self._check(self.L.vamd_feed_wrote(self.h, slot, nstreams, int(frames)))
        else:
            fr = np.ascontiguousarray(frames, dtype=np.int64)
            if fr.shape != (nstreams,):
                raise ValueError("frames must hold one length per stream")
            self._check(self.L.vamd_feed_wrote_v(self.h, slot, nstreams, _vp(fr.ctypes.data)))

    def packets(self, slot, copy=True):
        """Waits for the group.  -> dict: nstreams, nblocks, stream_start, offset, bits, granulepos, info (numpy views over
        the lane's pinned output arena, or copies), bytes, total_bytes, upload_ms, device_ms, total_ms."""
        r = _FeedResult()
        self._check(self.L.vamd_feed_packets(self.h, slot, C.byref(r)))
        nb, ns = int(r.nblocks), int(r.nstreams)

        def view(p, ct, n):
            if n == 0:
                return np.zeros(0, np.dtype(ct))
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,))
            return a.copy() if copy else a
        return {"nstreams": ns, "nblocks": nb, "stream_start": view(r.stream_start, C.c_int64, ns + 1),
                "offset": view(r.offset, C.c_int64, nb), "bits": view(r.bits, C.c_int32, nb),
                "granulepos": view(r.granulepos, C.c_int64, nb), "info": view(r.info, C.c_uint8, nb),
                "bytes": view(r.bytes, C.c_uint8, int(r.total_bytes)), "total_bytes": int(r.total_bytes),
                "upload_ms": r.upload_ms, "device_ms": r.device_ms, "total_ms": r.total_ms}

    def release(self, slot):
        self._check(self.L.vamd_feed_release(self.h, slot))

    def encode(self, pcm):
        """One group, synchronously: pcm [nstreams, frames, ch] of the feed's sample type (host), or a list of [frames_s, ch]
        arrays of unequal length.  -> per stream a list of (packet bytes, granulepos, W, e_o_s)."""
        if isinstance(pcm, (list, tuple)):
            parts = [np.ascontiguousarray(x, dtype=self.dtype) for x in pcm]
            ns, ch = len(parts), parts[0].shape[1]
            frames = np.array([x.shape[0] for x in parts], np.int64)
            flat = np.concatenate([x.reshape(-1) for x in parts])
        else:
            pcm = np.ascontiguousarray(pcm, dtype=self.dtype)
            ns, frames, ch = pcm.shape
            flat = pcm.reshape(-1)
        slot, buf = self.buffer(ch)
        try:
            buf[:flat.size] = flat
            self.wrote(slot, ns, frames)
            r = self.packets(slot)
        finally:
            try:
                self.release(slot)
            except VamdError:
                pass
        out = []
        for s in range(ns):
            row = []
            for k in range(int(r["stream_start"][s]), int(r["stream_start"][s + 1])):
                bits = int(r["bits"][k])
                o = int(r["offset"][k])
                data = bytes(r["bytes"][o:o + (bits + 7) // 8]) if bits >= 0 else None
                row.append((data, int(r["granulepos"][k]), int(r["info"][k]) & 1, (int(r["info"][k]) >> 1) & 1))
            out.append(row)
        return out


def envelope_marks(ret, first=0, marks=None):
    """Apply per-step flags to a mark array exactly as lib/envelope.c:241-258 does (host-side
    integer logic of the binding): step j = first + index."""
    n = first + len(ret) + 2
    if marks is None:
        marks = np.zeros(n, np.int32)
    elif len(marks) < n:
        marks = np.concatenate([marks, np.zeros(n - len(marks), np.int32)])
    for k, r in enumerate(ret):
        j = first + k
        marks[j + 2] = 0
        if r & 1:
            marks[j] = 1
            marks[j + 1] = 1
        if r & 2:
            marks[j] = 1
            if j > 0:
                marks[j - 1] = 1
    return marks
