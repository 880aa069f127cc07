#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection.csv files (one or more --pmc passes of the same command)
into one per-kernel table.  Dispatches are matched by order of appearance per kernel name.

    python tools/pmc_table.py out_dir1 [out_dir2 ...]  [--skip N leading dispatches per kernel]
"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = name.replace("bn::(anonymous namespace)::", "")
    return name[:name.index("(")] if "(" in name else name


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = OrderedDict()  # (kernel, grid, nth) -> {counter: value}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
            import sqlite3
            nth = defaultdict(int)
            keys = {}
            q = ("select dispatch_id, kernel_name, grid_size, workgroup_size, vgpr_count, lds_block_size, counter_name, value, start, end "
                 "from counters_collection order by dispatch_id")
            for did, kn, gs, wg, vg, lds, cn, val, st, en in sqlite3.connect(f).execute(q):
                key = keys.get(did)
                if key is None:
                    k = (short(kn), str(gs), str(wg))
                    key = keys[did] = k + (nth[k],)
                    nth[k] += 1
                e = rows.setdefault(key, {"dur": 0.0})
                e[cn] = e.get(cn, 0.0) + float(val)  # one row per counter instance (SE/XCC): sum
                e["dur"] = (en - st) / 1000.0
                e["vgpr"] = str(vg)
                e["lds"] = str(lds)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = defaultdict(dict)
            nth = defaultdict(int)
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    did = r["Dispatch_Id"]
                    key = seen[did].get("key")
                    if key is None:
                        k = (short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"])
                        key = k + (nth[k],)
                        nth[k] += 1
                        seen[did]["key"] = key
                    e = rows.setdefault(key, {"dur": 0.0})
                    e[r["Counter_Name"]] = float(r["Counter_Value"])
                    e["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
                    e["vgpr"] = r["VGPR_Count"]
                    e["lds"] = r["LDS_Block_Size"]
    # average over repeats of the same (kernel, grid)
    agg = OrderedDict()
    for (k, g, wg, n), e in rows.items():
        a = agg.setdefault((k, g, wg), defaultdict(float))
        a["_n"] += 1
        for c, v in e.items():
            if isinstance(v, float):
                a[c] += v
            else:
                a[c] = v
    names = sorted({c for a in agg.values() for c in a if c not in ("_n", "dur", "vgpr", "lds")})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid", "wg", "n", "vgpr", "lds", "dur_us"] + names)
    for (k, g, wg), a in agg.items():
        n = a["_n"]
        w.writerow([k, g, wg, str(int(n)), str(a["vgpr"]), str(a["lds"]), f"{a['dur'] / n:.1f}"] + [f"{a[c] / n:.0f}" for c in names])


if __name__ == "__main__":
    main()
