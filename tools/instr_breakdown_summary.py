"""Per-setting means of the counters collected around tools/instr_breakdown.py (3 launches per setting, in order)."""
import csv, glob, os, sys
SERIES = [(1, 0), (1, 1), (1, 2), (2, 2), (3, 2), (4, 2), (5, 2), (7, 2)]
rows = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if not name.startswith("tracking_step"):
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"kernel": name, "wg": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 64
print("launches", len(ids), rows[ids[0]]["kernel"], "workgroup", rows[ids[0]]["wg"], "grid", rows[ids[0]]["grid"])
names = [c for c in rows[ids[0]] if c not in ("kernel", "wg", "grid")]
prev = None
for s, setting in enumerate(SERIES):
    chunk = [rows[i] for i in ids[3 * s:3 * s + 3]]
    if not chunk:
        break
    mean = {c: sum(x.get(c, 0.0) for x in chunk) / len(chunk) / n_obj for c in names}
    line = "n_corr %d n_update %d: " % setting + "  ".join("%s %.0f" % (c.replace("SQ_", ""), mean[c]) for c in names)
    if prev:
        line += "   | delta VALU %.0f" % (mean.get("SQ_INSTS_VALU", 0) - prev.get("SQ_INSTS_VALU", 0))
    print(line)
    prev = mean
