import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for r in con.execute("select name, total_calls, average from top_kernels"):
    if r[0].startswith("k_") or r[0].startswith("void k_"):
        print("%-22s calls %d avg %.1f us" % (r[0].replace("void ", "").split("(")[0], r[1], r[2] / 1e3 if r[2] > 1e5 else r[2]))
