"""development aid: print the kernel timeline of a rocprofv3 --kernel-trace database (rocpd) for a window in the middle"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name,start,end,queue_id,stream_id from kernels order by start"))
t0 = rows[0][1]
mid = rows[int(len(rows) * float(sys.argv[3]) if len(sys.argv) > 3 else len(rows) // 2)][1]
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 14e6
busy_end = 0
idle = 0.0
for r in rows:
    if mid <= r[1] < mid + span:
        gap = (r[1] - busy_end) / 1e3 if busy_end and r[1] > busy_end else 0.0
        idle += gap
        nm = r[0].split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[-28:]
        print("%9.1f %8.1f q%s s%s %-28s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4], nm, ("idle %.0f" % gap) if gap > 20 else ""))
    busy_end = max(busy_end, r[2])
print("idle (no kernel on the chip) in the window: %.0f us of %.0f" % (idle, span / 1e3))
