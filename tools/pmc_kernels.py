#!/usr/bin/env python3
"""Per-kernel summary of one rocprofv3 --pmc pass (counter_collection.csv): for every kernel NAME (template
arguments kept) the mean per dispatch of each counter, the mean duration, and the derived shares

    python tools/pmc_kernels.py <rocprof output dir> [name filter]

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz); issue / stall / wait =
SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES; valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    n = name.replace("bn::(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n[:90]


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(dict)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if flt and flt not in k:
                    continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1000.0, r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count")
    for k in sorted(acc, key=lambda k: -sum(v[0] for v in disp[k].values())):
        n = len(disp[k])
        us = sum(v[0] for v in disp[k].values()) / n
        c = {name: val / n for name, val in acc[k].items()}
        any_d = next(iter(disp[k].values()))
        line = f"{k:90s} n={n:3d} us={us:7.1f} grid={any_d[1]} wg={any_d[2]} lds={any_d[3]} vgpr={any_d[4]}"
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            line += f" mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * us * 2400):.3f}"
        if wc:
            line += f" issue={c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} stall={c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} wait={c.get('SQ_WAIT_ANY', 0) / wc:.2f}"
        if "SQ_INSTS_VALU" in c:
            line += f" valu={c['SQ_INSTS_VALU']:.0f} salu={c.get('SQ_INSTS_SALU', 0):.0f} lds={c.get('SQ_INSTS_LDS', 0):.0f} vmem_rd={c.get('SQ_INSTS_VMEM_RD', 0):.0f} vmem_wr={c.get('SQ_INSTS_VMEM_WR', 0):.0f} waves={c.get('SQ_WAVES', 0):.0f}"
        if c.get("SQ_INSTS_MFMA"):
            # SQ_INSTS_VALU COUNTS the matrix instructions too (round 4, tools/mfma_valu_probe under --pmc: a matrix-only kernel reads
            # SQ_INSTS_VALU == SQ_INSTS_MFMA to 0.04 %): "vector" below is the difference; the valu/mfma ratio of the round 2 / 3 tables
            # was (vector + matrix) / matrix, i.e. one too high
            vec = c.get('SQ_INSTS_VALU', 0) - c['SQ_INSTS_MFMA']
            line += f" vector(valu-mfma)={vec:.0f} vector/mfma={vec / c['SQ_INSTS_MFMA']:.1f} mfma={c['SQ_INSTS_MFMA']:.0f}"
        if c.get("SQ_LDS_IDX_ACTIVE"):
            line += f" lds_conf={c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f} lds_idx_cycles={c['SQ_LDS_IDX_ACTIVE']:.0f} lds_conf_cycles={c.get('SQ_LDS_BANK_CONFLICT', 0):.0f}"
        for extra in ("FETCH_SIZE", "WRITE_SIZE"):
            if extra in c:
                line += f" {extra}={c[extra] / 1024:.1f}MiB"
        print(line)


if __name__ == "__main__":
    main()
