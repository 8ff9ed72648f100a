"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/collect_traffic.sh into HBM bytes per launch.

The counters are in KB.  On gfx950 FETCH_SIZE counts 64 B for each 128 B request of a wide stream
(MI355X_MICROARCH.md, HBM section), so the read side is doubled: an upper estimate for the scattered
gathers of these kernels."""
import csv
import glob
import json
import os
import sys


def counter_means(folder, counter):
    per_kernel = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            key = (name, int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))
            per_kernel.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
            per_kernel[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return per_kernel


def main(out_dir, target):
    fetch = counter_means(os.path.join(out_dir, "fetch"), "FETCH_SIZE")
    write = counter_means(os.path.join(out_dir, "write"), "WRITE_SIZE")
    kernels = {}
    objects = None
    for (name, groups), launches in fetch.items():
        if name not in ("tracking_step_kernel", "tracking_step_split_kernel", "region_histogram_kernel"):
            continue
        if len(launches) < 10:      # the bench's timed batch, not the one-off set-up launches
            continue
        w = write.get((name, groups), {})
        f_kb = sum(launches.values()) / len(launches)
        w_kb = sum(w.values()) / max(len(w), 1)
        kernels[name] = {"FETCH_SIZE_KB_mean": round(f_kb, 2), "FETCH_SIZE_launches": len(launches),
                         "WRITE_SIZE_KB_mean": round(w_kb, 2), "WRITE_SIZE_launches": len(w),
                         "hbm_bytes_per_launch_raw": int((f_kb + w_kb) * 1024),
                         "hbm_bytes_per_launch_corrected": int((2 * f_kb + w_kb) * 1024)}
        objects = groups // 4 if name == "tracking_step_split_kernel" else groups  # M3T_SPLIT_PARTS workgroups each
    json.dump({"command": "tools/collect_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE "
                          "(separate passes) -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline",
               "note": "read side doubled per MI355X_MICROARCH.md HBM section (gfx950 FETCH_SIZE counts 64 B per "
                       "128 B request on wide streams; upper estimate for the scattered gathers of these kernels)",
               "objects_per_launch": objects,
               "histogram_update_fused": "region_histogram_kernel" not in kernels,
               "kernels": kernels}, open(target, "w"), indent=1)
    print(open(target).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
