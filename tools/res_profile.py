"""The residue search and the packet assembly of a batch by themselves: HIP-event stage times (vamd_stage_ms) of the
PCM -> packets path over a batch of stereo long blocks, and the in-kernel phase stopwatch of k_residue / k_pack
(slots 8.. of the last row: residue classes + offsets, search, then k_pack's phases).

    python tools/res_profile.py [setup] [blocks]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vorbis_amd

setup = sys.argv[1] if len(sys.argv) > 1 else "44k_stereo_q4"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
an = vorbis_amd.Analyzer(vorbis_amd.default_setup_blob(setup), 0)
torch.manual_seed(0)
pcm = torch.rand((nb, an.channels, an.blocksizes[1]), device="cuda") - 0.5
pk = an.alloc_outputs(1, nb, ("ampmax_out", "packets", "packet_bits", "res_class", "res_entries", "res_count"))
for _ in range(2):
    an.analyze(pcm, outs=pk)
torch.cuda.synchronize()
an.profile(True)
N = 5
for _ in range(N):
    an.analyze(pcm, outs=pk)
ms, runs = an.stage_ms()
an.profile(False)
print("blocks %d, stage ms per batch:" % nb, {k: round(v / runs, 3) for k, v in ms.items()})
print("entries per block %.1f, packet bytes %.1f" % (float(pk["res_count"][:, 1].float().mean()), float(pk["packet_bits"].float().mean()) / 8))
an.debug_cycles(True)
an.analyze(pcm, outs=pk)
torch.cuda.synchronize()
c = an.debug_cycles(False, read=True)
GHZ = 2.4
row = c[4]
print("couple / residue / pack phase ticks per block (summed over the unit's waves), kcycles:", [round(float(x) / nb / 1e3, 2) for x in row])
