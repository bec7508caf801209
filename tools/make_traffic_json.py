"""profiles/spmv_traffic.json from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate passes) of bench.py.
usage: python tools/make_traffic_json.py [--merge] <git_head> c3d4:<fetch.db>:<write.db>[:<bench line.json>] [c3d10:...] [cpe8:...]
The file is stamped with a fingerprint of the machine code of the PCG / SpMV kernels in libfemcy_hip.so
(bench.kernel_object_sha) and, per workload, with the pattern sizes of the run (`config.layout` of the bench line given as
the fourth field): bench.py refuses the numbers once a kernel or the layout changes -- and only then (round 5 hashed
whole source files and lost the record to an unrelated edit of ctx.hpp)."""
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
            name = rows[0][0]
            # the launches of the timed steps only: a run may also hold shorter solves of the same kernel (the 100-iteration
            # solves of an `hbm_bound` record inside `--workload cpe8`), which must not dilute the per-launch average
            dmax = cur.execute("select max(duration) from counters_collection where counter_name=? and kernel_name=?",
                               (counter, name)).fetchone()[0]
            n, v, d = cur.execute("select count(*), avg(value), avg(duration) from counters_collection where counter_name=? "
                                  "and kernel_name=? and duration >= ?", (counter, name, 0.8 * dmax)).fetchone()
            return name, n, v, d / 1e3
    raise SystemExit(f"{db}: no k_pcg_persist / k_spmv dispatches with counter {counter}")


def main():
    head = sys.argv[1]
    path = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    keep = {}
    if "--merge" in sys.argv:                  # re-take some workloads, keep the others (same kernel sources only)
        sys.argv.remove("--merge")
        old = json.load(open(path))
        if old.get("kernel_object_sha") != bench.kernel_object_sha():
            raise SystemExit("--merge: the kernels changed since profiles/spmv_traffic.json was taken")
        keep = old.get("workloads", {})
    out = {"kernel_object_sha": bench.kernel_object_sha(), "git_head": head,
           "kernels": list(bench.TRAFFIC_KERNELS),
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, "
                         "HBM section) -> x2; WRITE_SIZE taken as reported (uncalibrated)",
           "note": "counters are L2 <-> fabric requests, Infinity-Cache hits included",
           "workloads": dict(keep)}
    for spec in sys.argv[2:]:
        parts = spec.split(":")
        wl, fdb, wdb = parts[:3]
        layout = None
        if len(parts) > 3:                                      # the JSON line of the profiled bench run
            for line in open(parts[3]):
                if line.startswith("{"):
                    layout = json.loads(line).get("config", {}).get("layout")
        kname, nf, fetch_kb, us_f = avg_counter(fdb, "FETCH_SIZE")
        _, nw, write_kb, us_w = avg_counter(wdb, "WRITE_SIZE")
        out["workloads"][wl] = {"kernel": kname, "dispatches": [nf, nw], "fetch_size_kb_reported": fetch_kb,
                                "write_size_kb_reported": write_kb, "avg_us_under_pmc": [us_f, us_w],
                                "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)), "layout": layout}
        frag = bench.kernel_symbol_fragment(kname)               # the ONE instantiation the passes ran: its machine code
        if frag:
            out["workloads"][wl]["kernel_sha"] = bench.kernel_object_sha(patterns=(frag,))
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
