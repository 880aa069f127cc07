#!/usr/bin/env python3
"""Condense one tools/profile_round.sh run into the files that are committed under profiles/.

    python tools/profile_summary.py gpurun_out/<tag> <tag> [v24|v30|perch] [batch]

Writes profiles/<tag>_bench_default.json, _bench_1stream.json, _kernel_stats_default.csv,
_kernel_stats_1stream.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc_traffic.json and refreshes
profiles/pmc_traffic.json (what bench.py quotes as `roofline.traffic`).

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE come from separate --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced reads, so reads are doubled:
    traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024        [bytes]
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def family(name):
    n = name.replace("bn::(anonymous namespace)::", "")
    for key, fam in (("gemm_mfma_kernel", "gemm_mfma_kernel"), ("gemm_splitk_kernel", "gemm_mfma_kernel"), ("gemm_dma_kernel", "gemm_mfma_kernel"), ("gemm_dma3_kernel", "gemm_mfma_kernel"), ("gemm_b3_kernel", "gemm_mfma_kernel"), ("frame_fold2q_kernel", "gemm_mfma_kernel"), ("frame_fold_kernel", "gemm_mfma_kernel"), ("frame_fold2_kernel", "gemm_mfma_kernel"), ("frame_fold2p_kernel", "gemm_mfma_kernel"), ("frame_foldh_kernel", "gemm_mfma_kernel"),
                     ("mbconv_", "mbconv_row_kernel"), ("mbmap_kernel", "mbmap_kernel"), ("mbmap_ws_kernel", "mbmap_kernel"),
                     ("dwconv_", "dwconv_kernel"), ("conv_small", "conv_direct_kernel"), ("conv_direct", "conv_direct_kernel"),
                     ("stft_kernel", "stft_kernel"), ("se_fc", "se_fc_kernel"), ("gap_partial", "gap_partial_kernel"), ("elt_", "elt_kernel"), ("reduce_", "reduce_kernel"), ("minmax_chunks", "reduce_kernel"),
                     ("topk", "topk_kernel")):
        if key in n:
            return fam
    return "other"


def counter_by_family(path, counter):
    out = defaultdict(lambda: [0.0, 0])
    seen = set()
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                fam = family(r["Kernel_Name"])
                out[fam][0] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    out[fam][1] += 1
    return out


def counters_by_family(path):
    """{family: {counter: sum over dispatches and counter instances, "_launches": n}} of one --pmc pass"""
    out = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                fam = family(r["Kernel_Name"])
                out[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen[fam]:
                    seen[fam].add(r["Dispatch_Id"])
                    out[fam]["_launches"] += 1
                    out[fam]["_dur_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    return out


MODEL, BATCH = "v24", 32  # set by main() from the command line
def suffix():
    return "" if MODEL == "v24" else "_" + MODEL


def mfma_summary(src, tag):
    """profiles/<tag>_pmc_mfma.json: per kernel family, per launch -- matrix-pipe busy fraction, LDS bank-conflict share,
    wait breakdown, effective clock.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs;
    GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back): cycles of the launch = GRBM / 8."""
    sq = counters_by_family(os.path.join(src, "pmc_sq"))
    inst = counters_by_family(os.path.join(src, "pmc_inst"))
    fams = {}
    for fam, c in sq.items():
        n = max(c["_launches"], 1.0)
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        wave = max(c["SQ_WAVE_CYCLES"], 1.0)
        e = {"launches_profiled": int(n), "avg_us": round(c["_dur_us"] / n, 2),
             "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024.0 * 2400.0 * c["_dur_us"], 1.0), 4),
             "mfma_busy_grbm": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024.0 * cyc, 1.0), 4),
             "grbm_cycles_per_us": round(cyc / max(c["_dur_us"], 1e-9), 1),
             "wave_cycles_share": {"active": round(c["SQ_ACTIVE_INST_ANY"] / wave, 3), "issue_stall": round(c["SQ_WAIT_INST_ANY"] / wave, 3),
                                   "waitcnt_or_barrier": round(c["SQ_WAIT_ANY"] / wave, 3)},
             "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 3)}
        i = inst.get(fam)
        if i:
            ni = max(i["_launches"], 1.0)
            e["per_launch"] = {k[3:].lower(): round(i[k] / ni) for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES") if k in i}
        fams[fam] = e
    doc = {"tag": tag, "batch": BATCH, "model": MODEL,
           "command": f"rocprofv3 --pmc <SQ counters + GRBM_GUI_ACTIVE> (separate passes, no trace domains) -- python3 tools/pmc_run.py {BATCH} 3 {MODEL}",
           "definitions": {"mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES (= 64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32, per SIMD) / (1024 SIMDs x launch duration x 2.4 GHz): "
                                        "the fraction of the matrix pipes' peak-clock cycles the launch kept busy",
                           "mfma_busy_grbm": "the same over GRBM_GUI_ACTIVE / 8; that quotient reads high on dispatches under ~0.3 ms "
                                             "(MI355X_MICROARCH.md, DVFS give-back), so this one reads LOW here -- kept for reference",
                           "wave_cycles_share": "SQ_ACTIVE_INST_ANY | SQ_WAIT_INST_ANY | SQ_WAIT_ANY over SQ_WAVE_CYCLES",
                           "lds_bank_conflict_share": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"},
           "families": fams}
    json.dump(doc, open(f"profiles/{tag}_pmc_mfma.json", "w"), indent=1)
    json.dump(doc, open(f"profiles/pmc_mfma{suffix()}.json", "w"), indent=1)
    return doc


def stats_by_family(path):
    fams = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                fam = family(r["Name"])
                fams[fam][0] += int(r["Calls"])
                fams[fam][1] += float(r["TotalDurationNs"])
    return {k: {"calls": v[0], "avg_us": round(v[1] / v[0] / 1000.0, 2), "total_ms": round(v[1] / 1e6, 3)} for k, v in fams.items() if v[0]}


def main():
    global MODEL, BATCH
    src, tag = sys.argv[1], sys.argv[2]
    MODEL = sys.argv[3] if len(sys.argv) > 3 else "v24"
    BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    os.makedirs("profiles", exist_ok=True)
    for name in ("bench_default", "bench_1stream"):
        line = [l for l in open(os.path.join(src, name + ".json")) if l.startswith("{")][-1]
        open(f"profiles/{tag}_{name}.json", "w").write(line)
    for name in ("default", "1stream"):
        f = glob.glob(os.path.join(src, f"trace_{name}", "**", "*kernel_stats.csv"), recursive=True)[0]
        shutil.copy(f, f"profiles/{tag}_kernel_stats_{name}.csv")
    fetch = counter_by_family(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = counter_by_family(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    traffic = {}
    for fam in sorted(set(fetch) | set(write)):
        fk, n = fetch.get(fam, [0.0, 0])
        wk, n2 = write.get(fam, [0.0, 0])
        n = max(n, n2, 1)
        traffic[fam] = {"launches_profiled": n, "fetch_size_kib_raw_per_launch": round(fk / n, 1), "write_size_kib_per_launch": round(wk / n, 1),
                        "hbm_bytes_per_launch": round((2.0 * fk + wk) * 1024.0 / n)}
    doc = {"tag": tag, "model": MODEL, "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/pmc_run.py {BATCH} 3 {MODEL}",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
           "batch": BATCH, "steps_profiled": 3, "families": traffic,
           "kernel_stats_1stream": stats_by_family(os.path.join(src, "trace_1stream")),
           "kernel_stats_default": stats_by_family(os.path.join(src, "trace_default"))}
    json.dump(doc, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    json.dump(doc, open(f"profiles/pmc_traffic{suffix()}.json", "w"), indent=1)
    print(json.dumps(doc["families"], indent=1))
    print(json.dumps(doc["kernel_stats_1stream"], indent=1))
    if os.path.isdir(os.path.join(src, "pmc_sq")):
        print(json.dumps(mfma_summary(src, tag)["families"], indent=1))
    for part in ("sq", "inst"):
        f = os.path.join(src, f"pmc_by_kernel_{part}.txt")
        if os.path.exists(f):
            shutil.copy(f, f"profiles/{tag}_pmc_by_kernel_{part}.txt")
    rec = os.path.join(src, "recording_24h.log")
    if os.path.exists(rec):
        shutil.copy(rec, f"profiles/{tag}_recording_24h.log")


if __name__ == "__main__":
    main()
