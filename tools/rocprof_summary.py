"""Summarise rocprofv3 rocpd .db outputs (kernel trace / PMC passes) as text for profiles/.
usage: python tools/rocprof_summary.py stats <results.db> | pmc <results.db> <COUNTER>"""
import sqlite3
import sys


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':<72} {'calls':>7} {'total_us':>12} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}")
    for name, n, s, a, mn, mx in rows:
        print(f"{name[:72]:<72} {n:>7} {s/1e3:>12.1f} {a/1e3:>9.3f} {mn/1e3:>9.3f} {mx/1e3:>9.3f} {100*s/tot:>6.2f}")


def pmc(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value), min(value), max(value), avg(duration) from "
                       "counters_collection where counter_name=? group by kernel_name order by sum(value) desc",
                       (counter,)).fetchall()
    print(f"{counter} per dispatch (KB as reported by rocprofv3; see MI355X_MICROARCH.md HBM section for the gfx950 "
          f"FETCH_SIZE x2 correction on wide streaming reads)")
    print(f"{'kernel':<72} {'calls':>6} {'avg':>12} {'min':>12} {'max':>12} {'avg_us':>9}")
    for name, n, a, mn, mx, d in rows:
        print(f"{name[:72]:<72} {n:>6} {a:>12.1f} {mn:>12.1f} {mx:>12.1f} {d/1e3:>9.2f}")


def pmc_all(db, pattern=None):
    """every counter of a PMC pass, per kernel: average per dispatch (summed over the counter's instances / XCDs as
    rocprofv3 stores them)"""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    last = None
    for name, ctr, n, a, d in rows:
        if pattern and pattern not in name:
            continue
        if name != last:
            print(f"\n{name[:110]}   ({n} dispatches, avg {d/1e3:.2f} us)")
            last = name
        print(f"    {ctr:<36} {a:>18.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmc_all":
        pmc_all(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        pmc(sys.argv[2], sys.argv[3])
