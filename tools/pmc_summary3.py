"""Per-launch means of the PMC passes of tools/collect_profiles.sh for every kernel of the timed batch, the ratios the
counters were collected for, and the HBM traffic block bench.py quotes (profiles/rNN_hbm_traffic_<profile>.json).
SQ_* cycle counters are per-SE sums of quad-cycles as rocprofv3 reports them; GRBM_GUI_ACTIVE comes out once per XCD
(summed here, divided by 8 below); FETCH_SIZE / WRITE_SIZE are in KB.  Corrections are MEASURED on this hardware in the
tracking kernels' own access patterns (profiles/rNN_counter_calibration.txt from tools/ubench_counters.hip, lines
`factor <pattern> <counter> <reported / bytes moved>`): the tracking kernels' HBM reads are first touches of frame rows
by lanes on different rows (cal_read_pixels: FETCH_SIZE reports 0.62 of the bytes) plus tables and model rows read in
order (cal_read_4B / cal_read_16B: 0.50); their writes are consecutive (cal_write_4B: 1.00).  The calibrated figure
divides FETCH_SIZE by the pixel-walk factor and leaves WRITE_SIZE alone; the bracket [raw, read side doubled] is
published beside it -- a sparse 4-byte gather is tallied at exactly 64 B per request (factor 1.00), a wide stream at
half its bytes (MI355X_MICROARCH.md §HBM), and everything these kernels do lies between the two.

  python tools/pmc_summary3.py <dir with one sub-directory per pass> <out.json> <profile name> "<command>"
"""
import csv
import glob
import json
import os
import sys

OBJECTS = {"rbot64": 64, "rbot4096": 4096, "ycb21": 21, "synth512": 512, "chain8": 8}
CONFIG = {"rbot64": "rbot64", "rbot4096": "rbot64", "ycb21": "ycb21", "synth512": "synth512", "chain8": "chain8"}


def calibration():
    """(FETCH_SIZE factor of the pixel-walk pattern, WRITE_SIZE factor of consecutive stores, file) from the newest
    profiles/r*_counter_calibration.txt; (0.5, 1.0, None) -- the guide's streaming-read figure -- without one"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for path in sorted(glob.glob(os.path.join(root, "r*_counter_calibration.txt")), reverse=True):
        f = {}
        for line in open(path):
            p = line.split()
            if len(p) == 4 and p[0] == "factor":
                f[(p[1], p[2])] = float(p[3])
        if ("cal_read_pixels", "FETCH_SIZE") in f:
            return f[("cal_read_pixels", "FETCH_SIZE")], f.get(("cal_write_4B", "WRITE_SIZE"), 1.0), os.path.basename(path)
    return 0.5, 1.0, None


def per_kernel(folder):
    per = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            if not (name.startswith("tracking_step") or name == "region_histogram_kernel"):
                continue
            key = (name, int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1), int(row["Workgroup_Size"]),
                   int(row.get("VGPR_Count", 0) or 0) + int(row.get("Accum_VGPR_Count", 0) or 0),
                   int(row.get("LDS_Block_Size", 0) or 0), int(row.get("Scratch_Size", 0) or 0))
            d = per.setdefault(key, {}).setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    out = {}
    for key, counters in per.items():
        n = max(len(v) for v in counters.values())
        if n < 5:  # the timed batch, not one-off set-up launches
            continue
        out[key] = {c: sum(v.values()) / len(v) for c, v in counters.items()}
        out[key]["_launches"] = n
    return out


def ratios_of(m):
    g = m.get
    r = {}
    if g("SQ_WAVE_CYCLES"):
        r["wave_cycles_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
        r["wave_cycles_issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)"] = g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
        r["wave_cycles_issuing_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)"] = g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    gui = g("GRBM_GUI_ACTIVE", 0) / 8.0
    if gui:
        r["kernel_cycles (GRBM_GUI_ACTIVE / 8 XCDs)"] = gui
    if g("SQ_ACTIVE_INST_VALU") and gui:
        r["valu_busy_pct (100 x SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / kernel cycles)"] = \
            100.0 * g("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / gui
    if g("SQ_WAVE_CYCLES") and gui:
        r["mean_resident_waves_per_cu (SQ_WAVE_CYCLES x 4 / kernel cycles / 256)"] = g("SQ_WAVE_CYCLES") * 4.0 / gui / 256.0
    if g("SQ_WAVES") and g("SQ_INSTS_VALU") is not None:
        r["valu_instructions_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    if g("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_frac (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"] = g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")
    if g("TCC_HIT_sum") is not None and (g("TCC_HIT_sum", 0) + g("TCC_MISS_sum", 0)):
        r["l2_hit_frac (TCC_HIT / (TCC_HIT + TCC_MISS))"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TA_TA_BUSY_sum") and gui:
        r["ta_busy_frac_per_cu (TA_TA_BUSY_sum / (256 x kernel cycles))"] = g("TA_TA_BUSY_sum") / (256.0 * gui)
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and gui:
        r["l1_accesses_per_cycle_per_cu"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") / (256.0 * gui)
    if g("TCP_PENDING_STALL_CYCLES_sum") and gui:
        r["l1_pending_stall_frac_per_cu"] = g("TCP_PENDING_STALL_CYCLES_sum") / (256.0 * gui)
    return r


def main(out_dir, target, profile, command):
    kernels = {}
    for p in ("sq1", "sq2", "ta", "tcc", "fetch", "write"):
        for key, c in per_kernel(os.path.join(out_dir, p)).items():
            k = kernels.setdefault(key[0], {"workgroups": key[1], "threads": key[2], "vgprs": key[3],
                                            "lds_bytes": key[4], "scratch_bytes": key[5], "per_launch_means": {}})
            k["per_launch_means"].update({n: v for n, v in c.items()})
    f_read, f_write, cal_file = calibration()
    traffic = {"command": command, "profile": profile, "config": CONFIG.get(profile, profile),
               "objects_per_launch": OBJECTS.get(profile), "kernels": {},
               "correction": {"FETCH_SIZE": round(1.0 / f_read, 4), "WRITE_SIZE": round(1.0 / f_write, 4),
                              "how": ("calibrated, profiles/%s: FETCH_SIZE reports %.3f of the bytes of the pixel-walk "
                                      "pattern (a lane per image row), WRITE_SIZE %.3f of consecutive stores" %
                                      (cal_file, f_read, f_write)) if cal_file else
                                     "uncalibrated: the guide's streaming-read factor (FETCH_SIZE x 2)"},
               "note": "FETCH_SIZE and WRITE_SIZE (KB) in separate rocprofv3 --pmc passes, mean over the launches of the "
                       "timed region; hbm_bytes_per_launch_corrected applies `correction`, the bracket is [raw, read "
                       "side doubled] (sparse gathers are tallied at 1.00, wide streams at 0.50 of their bytes)"}
    for name, k in kernels.items():
        m = k["per_launch_means"]
        k["ratios"] = ratios_of(m)
        if m.get("FETCH_SIZE") is not None and m.get("WRITE_SIZE") is not None:
            k["hbm"] = {"FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
                        "hbm_bytes_per_launch_raw": int((m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024),
                        "hbm_bytes_per_launch_corrected": int((m["FETCH_SIZE"] / f_read + m["WRITE_SIZE"] / f_write) * 1024),
                        "hbm_bytes_per_launch_bracket": [int((m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024),
                                                         int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)]}
            traffic["kernels"][name] = dict(k["hbm"])
    # fused = the histogram update rides in the tracking launch: then region_histogram_kernel only runs for
    # StartModalities (and in bench.py's unfused-buckets leg, which profiling runs skip: --no-buckets), i.e. far less
    # often than the tracking kernel
    tracking = max((k["per_launch_means"].get("_launches", 0) for n, k in kernels.items() if n.startswith("tracking_step")),
                   default=0)
    hist = kernels.get("region_histogram_kernel", {}).get("per_launch_means", {}).get("_launches", 0)
    traffic["histogram_update_fused"] = hist < 0.5 * tracking
    traffic["launches"] = {"tracking": tracking, "region_histogram_kernel": hist}
    json.dump({"command": command, "profile": profile, "kernels": kernels}, open(target, "w"), indent=1)
    json.dump(traffic, open(os.path.join(os.path.dirname(os.path.abspath(target)), "hbm_traffic_%s.json" % profile), "w"), indent=1)
    for name, k in kernels.items():
        print(name, k["workgroups"], "x", k["threads"], "vgprs", k["vgprs"], "lds", k["lds_bytes"])
        print(json.dumps(k["ratios"], indent=1))
        print(json.dumps(k.get("hbm")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
