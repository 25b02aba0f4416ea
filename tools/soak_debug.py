"""Scratch: regenerate one soak configuration's short-block batch exactly as tests/soak_lib.run(NB) does and compare the
GPU's taps of given blocks with the reference's, stage by stage; also the same block analysed alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vorbis_amd
from oracle import ref
from tests import soak_lib, checker
cfg = (6, 44100, 0.3, True); NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
blocks = [int(v) for v in sys.argv[2:]] or [782, 1998]
ch, rate, q, coupled = cfg
e = ref.RefEncoder(ch, rate, q, coupled=coupled)
an = vorbis_amd.Analyzer(e.pack_setup(), 0)
rng = np.random.default_rng(hash((ch, rate, int(q * 10))) & 0xffff)
for W in (1, 0):
    n = e.blocksize(W); nb = NB if W else NB // 3
    x = soak_lib.signals(rng, nb, ch, n)
    lW = rng.integers(0, 2, nb).astype(np.int32) * W; nW = rng.integers(0, 2, nb).astype(np.int32) * W
    bt = (rng.integers(0, 2, nb)).astype(np.int32)
    amp_in = np.where(rng.random(nb) < 0.5, -9999.0, rng.uniform(-60, 0, nb)).astype(np.float32)
    if W: continue
    want = ("mdct_raw", "logfft", "noise", "tone", "logmask", "mdct", "posts", "post_valid", "ilogmask", "iwork", "nonzero", "local_ampmax", "ampmax_out")
    o = an.analyze(torch.from_numpy(x).cuda(), W=W, lW=lW, nW=nW, blocktype=bt, ampmax_in=amp_in, want=want)
    torch.cuda.synchronize()
    host = {k: v.cpu().numpy() for k, v in o.items()}
    for k in blocks:
        a = e.tap_block(x[k], int(lW[k]), W, int(nW[k]), int(bt[k]), float(amp_in[k]))
        g = {kk: host[kk][k] for kk in host}
        print("block", k, "in the batch: differing tensors", checker.compare_block(a, g, 65, verbose=True))
        o1 = an.analyze(torch.from_numpy(x[k:k + 1]).cuda(), W=W, lW=lW[k:k + 1], nW=nW[k:k + 1], blocktype=bt[k:k + 1], ampmax_in=amp_in[k:k + 1], want=want)
        torch.cuda.synchronize()
        g1 = {kk: v.cpu().numpy()[0] for kk, v in o1.items()}
        print("block", k, "alone: differing tensors", checker.compare_block(a, g1, 65, verbose=True))
        np.savez("/root/repo/gpurun_out/soak_block_%d.npz" % k, pcm=x[k], lW=lW[k], nW=nW[k], bt=bt[k], amp_in=amp_in[k])
