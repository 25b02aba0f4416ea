"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel average of each counter, per wave."""
import sqlite3, sys, collections
for f in sys.argv[1:]:
    con = sqlite3.connect(f)
    rows = con.execute("select kernel_name, counter_name, sum(value), count(*), max(grid_size), max(workgroup_size) from counters_collection group by kernel_name, counter_name").fetchall()
    d = collections.defaultdict(dict)
    waves = {}
    for k, c, v, n, g, wg in rows:
        if k.startswith("void "):   # instantiated kernels print as "void k_transform<11>(...)"
            k = k[5:]
        if not k.startswith("k_"):
            continue
        kk = k.split("(")[0]
        d[kk][c] = v / n
        waves[kk] = g / 64
    for k, v in d.items():
        w = waves[k]
        print(k, "waves", int(w), {c: round(x / w, 1) for c, x in sorted(v.items())})
