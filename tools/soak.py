"""The random-signal soak (tests/soak_lib.py) at any size from the command line; the test-suite runs it with
>= 10 000 blocks as tests/test_gpu_soak.py.

    python tools/soak.py [blocks per configuration, default 300]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import soak_lib

total, bad = soak_lib.run(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
print("SOAK", "FAILED" if bad else "OK", total, "blocks", bad, "mismatches")
sys.exit(1 if bad else 0)
