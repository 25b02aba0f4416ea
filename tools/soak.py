"""The random-signal soak (tests/soak_lib.py) at any size from the command line; the test-suite runs it with
>= 10 000 blocks as tests/test_gpu_soak.py.

    python tools/soak.py [blocks per configuration, default 300] [log file]

Every run writes its log -- source hash of the kernels it ran, every configuration's running totals, every mismatch,
the verdict -- to the log file (default gpurun_out/soak.txt); a full-size run's log is committed as
profiles/rNN_soak.txt, and the last one before a round ends must be a passing one at HEAD's source hash.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tests import soak_lib

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "soak.txt")
os.makedirs(os.path.dirname(path), exist_ok=True)
out = open(path, "w")


def log(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    out.write(line + "\n")
    out.flush()


log("# tools/soak.py %d : random-signal soak of the batch path against the reference's real vorbis_analysis()" % nb)
log("# source_hash", bench.source_hash(), " started", time.strftime("%Y-%m-%d %H:%M:%S"))
log("# signal kinds:", soak_lib.NKINDS, "(8-11: denormals, signed zeros, +30..+60 dB noise, +86 dB impulses; 12-13: noise +60..+90 dB, a sine +70..+85 dB)")
total, bad = soak_lib.run(nb, log=log)
checks, hbad = soak_lib.run_hostile(48, log=log)
log("SOAK", "FAILED" if bad else "OK", total, "blocks", bad, "mismatches")
log("HOSTILE", "FAILED" if hbad else "OK", checks, "checks", hbad, "failures")
sys.exit(1 if (bad or hbad) else 0)
