"""Turn rocprofv3 rocpd sqlite outputs into the text summaries committed under profiles/."""
import collections
import sqlite3
import sys


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("%-46s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for n, c, t, a, p in rows:
        n = n.replace("(anonymous namespace)::", "")
        print("%-46s %6d %14.0f %12.0f %6.2f%%" % (n.split("(")[0][:46], c, t, a, p))
    try:  # the `kernels` view of rocpd: one row per dispatch with its code-object resources
        r = con.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                        "max(workgroup_x), max(grid_x) from kernels group by name").fetchall()
        print("\n# code-object resources as rocpd's `kernels` view reports them (vgpr: half the allocated registers per lane --")
        print("# k_floor's 64 show as 32, k_transform's 232 as 116; tools/kernel_resources.py prints the compiler's own figures)")
        print("%-46s %6s %6s %6s %9s %8s %6s %10s" % ("kernel", "vgpr/2", "agpr", "sgpr", "lds_B", "scratch", "wg", "grid"))
        for n, v, a, sg, l, sc, w, g in r:
            n = n.replace("(anonymous namespace)::", "")
            nm = n[5:] if n.startswith("void ") else n
            if nm.startswith("k_"):
                print("%-46s %6s %6s %6s %9s %8s %6s %10s" % (nm.split("(")[0][:46], v, a, sg, l, sc, w, g))
    except Exception as e:  # view / column names differ between rocprofv3 builds
        print("(kernel resource table unavailable: %s)" % e)


def pmc(db):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    d = collections.defaultdict(dict)
    for k, c, v, n in rows:
        d[k.split("(")[0][:60]][c] = (v / n, n)
    print("%-60s %-14s %16s %8s" % ("kernel", "counter", "avg_per_dispatch", "n"))
    for k, v in d.items():
        for c, (x, n) in sorted(v.items()):
            print("%-60s %-14s %16.1f %8d" % (k, c, x, n))


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    kernel_stats(db) if mode == "kt" else pmc(db)
