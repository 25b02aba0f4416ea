"""One call's kernels on a time line, from a rocprofv3 --kernel-trace database: start relative to the call's first kernel,
duration, and the gap since the previous kernel on the time line ended (tools/kt_block.sh).  The call is picked as the
last complete run of `per_call` dispatches."""
import sqlite3, sys
db, first = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_transform"
con = sqlite3.connect(db)
try:
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
except Exception as e:
    print("kernels view unavailable:", e, [r[0] for r in con.execute("select name from sqlite_master")][:40])
    sys.exit(0)
rows = [(n[5:] if n.startswith("void ") else n, s, e) for n, s, e in rows]
rows = [(n.split("(")[0], s, e) for n, s, e in rows if n.startswith("k_")]
# calls begin at a kernel whose name starts with `first`; take the median-length call among the last 50
starts = [i for i, r in enumerate(rows) if r[0].startswith(first)]
calls = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])][-50:]
if not calls:
    sys.exit("no calls found")
calls.sort(key=lambda c: c[-1][2] - c[0][1])
call = calls[len(calls) // 2]
t0 = call[0][1]
print("%-28s %9s %9s %9s" % ("kernel", "start_us", "dur_us", "gap_us"))
busy_end = t0
for n, s, e in call:
    print("%-28s %9.1f %9.1f %9.1f" % (n[:28], (s - t0) / 1e3, (e - s) / 1e3, (s - busy_end) / 1e3))
    busy_end = max(busy_end, e)
print("first kernel start -> last kernel end: %.1f us (median of %d calls)" % ((busy_end - t0) / 1e3, len(calls)))
