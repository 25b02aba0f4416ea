"""tests/checker.py -- picks the strongest CPU checker available and compares tensors.

Order of preference:
  1. "reference": oracle/_ref/libvorbis_ref.so -- the unmodified reference sources compiled in place
     (travels to the GPU box as a prebuilt .so; rebuilt here when /root/reference exists);
  2. "port": oracle/libvorbis_port.so -- the from-scratch C restatement under oracle/port/, itself
     pinned against (1) and against tests/golden by the CPU test-suite.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETUPS = {"44k_stereo_q4": (2, 44100, 0.4), "44k_stereo_q9": (2, 44100, 0.9), "44k_stereo_q1": (2, 44100, 0.1),
          "44k_mono_q5": (1, 44100, 0.5)}

# the 5.1 layout (two submaps, four coupling steps; q 0.3 keeps coupling and noise normalisation both live).
# Kept apart from SETUPS (the fixture carries the decisions and the packet, not the float taps).
SURROUND = {"44k_51_q3": (6, 44100, 0.3)}

FLOAT_KEYS = ("mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct", "local_ampmax")
INT_KEYS = ("post_valid", "ilogmask", "iwork", "nonzero")


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


class Checker:
    def __init__(self, setup_name, prefer=None):
        from oracle import ref
        self.setup_name = setup_name
        ch, rate, q = SETUPS[setup_name]
        self.kind = None
        if prefer in (None, "reference") and ref.available():
            self.enc = ref.RefEncoder(ch, rate, q)
            self.kind = "reference"
        else:
            from oracle import port
            blob = np.fromfile(os.path.join(ROOT, "vorbis_amd", "data", "setup_%s.bin" % setup_name), dtype=np.uint8)
            self.enc = port.PortEncoder(blob)
            self.kind = "port"

    def tap_block(self, pcm, lW=1, W=1, nW=1, blocktype=1, ampmax_in=-9999.0):
        return self.enc.tap_block(pcm, lW, W, nW, blocktype, ampmax_in)

    def mdct_forward(self, W, x):
        return self.enc.mdct_forward(W, x)


def compare_block(ref, got, nposts, keys=None, verbose=False):
    """Bit-exact comparison of one block's tensors.  Returns the number of differing tensors."""
    nbad = 0
    for k in (FLOAT_KEYS + INT_KEYS if keys is None else keys):
        if k not in got or k not in ref:
            continue
        r, g = _bits(np.asarray(ref[k])), _bits(np.asarray(got[k]))
        if r.shape != g.shape or not np.array_equal(r, g):
            nbad += 1
            if verbose:
                d = np.nonzero(r.ravel() != g.ravel())[0] if r.shape == g.shape else []
                print("  MISMATCH %s: %d elements differ, first at %s" % (k, len(d), list(d[:5])))
    if "posts" in got and "posts" in ref:
        if not np.array_equal(np.asarray(ref["posts"])[:, :nposts], np.asarray(got["posts"])[:, :nposts]):
            nbad += 1
            if verbose:
                print("  MISMATCH posts")
    if "ampmax_out" in got and "ampmax_out" in ref:
        if np.float32(ref["ampmax_out"]) != np.float32(got["ampmax_out"]):
            nbad += 1
            if verbose:
                print("  MISMATCH ampmax_out", ref["ampmax_out"], got["ampmax_out"])
    return nbad
