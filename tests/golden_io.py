"""Loader for tests/golden/blocks_<setup>.npz (written by tools/make_golden.py from the reference)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAPS = ("windowed", "mdct_raw", "fft_packed", "logfft", "logmdct", "noise", "tone", "logmask", "mdct", "posts",
        "post_valid", "ilogmask", "iwork", "nonzero", "local_ampmax", "res_class", "res_entries")


def load(setup_name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "blocks_%s.npz" % setup_name))
    nb = int(z["nblocks"][0])
    blocks = []
    for i in range(nb):
        lW, W, nW, bt = [int(v) for v in z["b%d_desc" % i]]
        b = dict(lW=lW, W=W, nW=nW, blocktype=bt, ampmax_in=float(z["b%d_ampmax" % i][0]),
                 ampmax_out=float(z["b%d_ampmax" % i][1]), pcm=z["b%d_pcm" % i], packet=z["b%d_packet" % i].tobytes())
        for k in TAPS:
            if "b%d_%s" % (i, k) in z.files:   # (six-channel fixtures carry the decisions and the packet only)
                b[k] = z["b%d_%s" % (i, k)]
        blocks.append(b)
    fn = {k: z[k] for k in ("mdct0_in", "mdct0_out", "drft0_out", "mdct1_in", "mdct1_out", "drft1_out")}
    return blocks, [int(v) for v in z["posts"]], fn
