import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for r in con.execute("select name, total_calls, average from top_kernels"):
    if r[0].startswith("k_"):
        print("%-16s calls %d avg %.1f us" % (r[0].split("(")[0], r[1], r[2] / 1e3 if r[2] > 1e5 else r[2]))
