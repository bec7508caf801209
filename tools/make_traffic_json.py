"""profiles/spmv_traffic.json from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate passes) of bench.py.
usage: python tools/make_traffic_json.py <git_head> c3d4:<fetch.db>:<write.db> [c3d10:<fetch.db>:<write.db>]
The file is stamped with a fingerprint of the SpMV kernel sources (bench.kernel_source_sha): bench.py refuses the
numbers once those sources change."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def avg_counter(db, counter):
    """the dominant kernel of the run: the persistent PCG when the solves went through it, else the SpMV"""
    cur = sqlite3.connect(db).cursor()
    for like in ("%k_pcg_persist%", "%k_spmv%"):
        rows = cur.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where "
                           "counter_name=? and kernel_name like ? group by kernel_name", (counter, like)).fetchall()
        if rows:
            rows.sort(key=lambda r: -r[1])
            name, n, v, d = rows[0]
            return name, n, v, d / 1e3
    raise SystemExit(f"{db}: no k_pcg_persist / k_spmv dispatches with counter {counter}")


def main():
    head = sys.argv[1]
    out = {"kernel_source_sha": bench.kernel_source_sha(), "git_head": head,
           "sources": list(bench.TRAFFIC_SOURCES),
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, "
                         "HBM section) -> x2; WRITE_SIZE taken as reported (uncalibrated)",
           "note": "counters are L2 <-> fabric requests, Infinity-Cache hits included",
           "workloads": {}}
    for spec in sys.argv[2:]:
        wl, fdb, wdb = spec.split(":")
        kname, nf, fetch_kb, us_f = avg_counter(fdb, "FETCH_SIZE")
        _, nw, write_kb, us_w = avg_counter(wdb, "WRITE_SIZE")
        out["workloads"][wl] = {"kernel": kname, "dispatches": [nf, nw], "fetch_size_kb_reported": fetch_kb,
                                "write_size_kb_reported": write_kb, "avg_us_under_pmc": [us_f, us_w],
                                "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024))}
    json.dump(out, open(os.path.join(ROOT, "profiles", "spmv_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
