#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time stats and PMC counter
means per dispatch.  usage: rocpd_summary.py <dir-with-db-files> [kernel-substr]"""
import sqlite3, sys, glob, os, json
root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "gmm"
out = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    con = sqlite3.connect(f)
    name = os.path.relpath(f, root)
    out[name] = {}
    try:
        rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
        out[name]["kernels"] = [{"name": r[0][:90], "calls": r[1], "avg_us": r[2] / 1e3, "min_us": r[3] / 1e3, "max_us": r[4] / 1e3, "total_us": r[5] / 1e3} for r in rows[:8]]
    except Exception as e:
        out[name]["kernels_error"] = str(e)
    if os.environ.get("JAMD_BY_GRID"):
        # the same per (kernel, grid size): one kernel name can cover launches of very different sizes in one process
        # (the default bench line: gmm_tile at 64 000 frames for C2 and at 360 000 - 727 000 frames inside the e2e steps)
        try:
            cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
            gcol = next((c for c in cols if "grid" in c.lower() and c.lower().endswith("x")), None) or next((c for c in cols if "grid" in c.lower()), None)
            if gcol:
                rows = con.execute(f"select name, {gcol}, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name, {gcol} order by sum(end-start) desc").fetchall()
                out[name]["kernels_by_grid"] = [{"name": r[0][:90], "grid": r[1], "calls": r[2], "avg_us": r[3] / 1e3, "min_us": r[4] / 1e3, "max_us": r[5] / 1e3} for r in rows[:24] if sub in r[0]]
            else:
                out[name]["kernels_by_grid_error"] = "no grid column among " + ",".join(cols)
        except Exception as e:
            out[name]["kernels_by_grid_error"] = str(e)
    try:
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        pm = {}
        for k, c, v, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if sub in k:
                pm.setdefault(k[:60], {})[c] = v
        if pm:
            out[name]["pmc_avg_per_dispatch"] = pm
    except Exception as e:
        out[name]["pmc_error"] = str(e) + " cols=" + ",".join(cols)
print(json.dumps(out, indent=1))
