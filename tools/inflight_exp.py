"""Scratch: K contexts in flight on one GPU (their steps issued round-robin from one host thread; each context has its own
streams, so one's latency-bound stretches -- the detector's walk, the plan, kernel tails -- run beside the other's
analysis).  Usage: python tools/inflight_exp.py c4|c5 [K...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vorbis_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
ks = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
a = bench.parse(["--workload", wl, "--no-cpu-baseline", "--no-workloads"])
dev = torch.device("cuda", 0)
blob = vorbis_amd.default_setup_blob(a.setup or ("44k_stereo_q9" if wl == "c5" else "44k_stereo_q4"))
mk = bench.StreamRunner if wl == "c5" else bench.GpuRunner
for K in ks:
    Rs = [mk(a, blob, dev, 0, 1) for _ in range(K)]
    steps = 12 * K if wl == "c5" else 30 * K
    for r in range(3):
        for R in Rs:
            R.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            Rs[i % K].step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s in flight %d: %.3f ms per step, %.2f M units/s" % (wl, K, dt / steps * 1e3, Rs[0].units * steps / dt / 1e6), flush=True)
    del Rs
    torch.cuda.empty_cache()
