"""Regenerates floor1_db_table.h: parses the spec's floor1_inverse_dB_table literals
from a libvorbis checkout (lib/floor1.c) and emits their fp32 bit patterns."""
import re
import sys
import numpy as np

src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/lib/floor1.c").read()
m = re.search(r"FLOOR1_fromdB_LOOKUP\[256\]=\{(.*?)\};", src, re.S)
vals = [v.strip() for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
bits = np.array([float(v.rstrip("Ff")) for v in vals], dtype=np.float32).view(np.uint32)
for r in range(0, 256, 8):
    print("  " + ", ".join("0x%08xu" % b for b in bits[r:r + 8]) + (", \\" if r < 248 else ""))
