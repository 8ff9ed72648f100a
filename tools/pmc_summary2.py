"""Per-launch means of the PMC passes of tools/collect_pmc.sh for the tracking kernel of the timed batch, plus the
ratios the counters were collected for.  SQ_* cycle counters are per-SE sums of quad-cycles as rocprofv3 reports them;
FETCH_SIZE / WRITE_SIZE are in KB, the read side is doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B
request of a wide stream: an upper estimate for scattered gathers)."""
import csv
import glob
import json
import os
import sys

KERNELS = ("tracking_step_split_kernel", "tracking_step_kernel", "tracking_step_lds_kernel")


def means(folder):
    per = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            if name not in KERNELS:
                continue
            key = (name, int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1), int(row["Workgroup_Size"]),
                   int(row.get("VGPR_Count", 0) or 0), int(row.get("LDS_Block_Size", 0) or 0))
            d = per.setdefault(key, {}).setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    out = {}
    for key, counters in per.items():
        n = max(len(v) for v in counters.values())
        if n < 10:  # the bench's timed batch, not one-off set-up launches
            continue
        out[key] = {c: sum(v.values()) / len(v) for c, v in counters.items()}
        out[key]["_launches"] = n
    return out


def main(out_dir, target, config):
    merged, shape = {}, None
    for p in ("sq1", "sq2", "ta", "tcc", "fetch", "write"):
        for key, c in means(os.path.join(out_dir, p)).items():
            shape = key
            for k, v in c.items():
                merged[k] = v
    g = merged.get
    ratios = {}
    if g("SQ_WAVE_CYCLES"):
        ratios["wave_cycles_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
        ratios["wave_cycles_issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)"] = g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
        ratios["wave_cycles_issuing_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)"] = g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    # GRBM_GUI_ACTIVE comes out as the sum over the 8 XCDs (rocprofv3 prints one row per XCC, means() adds them up):
    # one XCD's count is an eighth -- 403 k cycles for the 161 us launch of rbot64
    gui = g("GRBM_GUI_ACTIVE", 0) / 8.0
    if g("SQ_ACTIVE_INST_VALU") and gui:
        # rocprof's VALUBusy: 100 x SQ_ACTIVE_INST_VALU x 4 / (number of SIMDs = 4 x 256) / GRBM_GUI_ACTIVE
        ratios["valu_busy_pct (100 x SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs))"] = \
            100.0 * g("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / gui
    if g("SQ_WAVE_CYCLES") and gui:
        # SQ_WAVE_CYCLES counts quad-cycles summed over all waves
        ratios["mean_resident_waves_per_cu (SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8) / 256)"] = \
            g("SQ_WAVE_CYCLES") * 4.0 / gui / 256.0
    if g("SQ_WAVES") and g("SQ_INSTS_VALU") is not None:
        ratios["valu_instructions_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    if g("SQ_LDS_IDX_ACTIVE"):
        ratios["lds_bank_conflict_frac (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"] = g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")
    if g("TCC_HIT_sum") is not None and (g("TCC_HIT_sum", 0) + g("TCC_MISS_sum", 0)):
        ratios["l2_hit_frac (TCC_HIT / (TCC_HIT + TCC_MISS))"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TA_TA_BUSY_sum") and gui:
        ratios["ta_busy_frac_per_cu (TA_TA_BUSY_sum / (256 x GRBM_GUI_ACTIVE / 8))"] = g("TA_TA_BUSY_sum") / (256.0 * gui)
    if g("TCP_TCP_TA_DATA_STALL_CYCLES_sum") and gui:
        ratios["tcp_ta_data_stall_frac_per_cu"] = g("TCP_TCP_TA_DATA_STALL_CYCLES_sum") / (256.0 * gui)
    hbm = None
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        hbm = {"FETCH_SIZE_KB": g("FETCH_SIZE"), "WRITE_SIZE_KB": g("WRITE_SIZE"),
               "hbm_bytes_per_launch_raw": int((g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024),
               "hbm_bytes_per_launch_corrected": int((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)}
    doc = {"command": "tools/collect_pmc.sh %s: rocprofv3 --kernel-trace --pmc <pass> -- python bench.py --config %s "
                      "--steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeats 1 (six passes)" % (config, config),
           "kernel": None if shape is None else {"name": shape[0], "workgroups": shape[1], "threads": shape[2],
                                                  "vgprs": shape[3], "lds_bytes": shape[4]},
           "per_launch_means": {k: v for k, v in sorted(merged.items())}, "ratios": ratios, "hbm": hbm}
    json.dump(doc, open(target, "w"), indent=1)
    print(json.dumps(doc["ratios"], indent=1))
    print(json.dumps(hbm))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rbot64")
