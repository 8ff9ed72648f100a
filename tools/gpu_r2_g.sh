#!/bin/bash
# instruction breakdown (unsplit 256 threads, split default) + sq passes at 512 objects
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2g
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
M3T_HIP_NO_SPLIT=1 M3T_HIP_THREADS=256 timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/unsplit256" -- python $REPO/tools/instr_breakdown.py 64 > "$OUT/unsplit256.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/split" -- python $REPO/tools/instr_breakdown.py 64 > "$OUT/split.log" 2>&1
M3T_HIP_NO_SPLIT=1 M3T_HIP_THREADS=256 timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/ycb_unsplit256" -- python $REPO/tools/instr_breakdown.py 21 ycb > "$OUT/ycb_unsplit256.log" 2>&1
for m in unsplit256 split; do python $REPO/tools/instr_breakdown_summary.py "$OUT/$m" 64 > "$OUT/$m.txt" 2>&1; done
python $REPO/tools/instr_breakdown_summary.py "$OUT/ycb_unsplit256" 21 > "$OUT/ycb_unsplit256.txt" 2>&1
CMD="python $REPO/bench.py --config synth512 --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --repeats 1"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$OUT/s512_sq1" -- $CMD > "$OUT/s512_sq1.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/s512_sq2" -- $CMD > "$OUT/s512_sq2.log" 2>&1
python - <<PY > "$OUT/s512.txt" 2>&1
import csv, glob
for p in ("s512_sq1", "s512_sq2"):
    per = {}
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0]
            d = per.setdefault((name, r["Grid_Size"], r["Workgroup_Size"]), {}).setdefault(r["Counter_Name"], {})
            d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for key, c in per.items():
        n = max(len(v) for v in c.values())
        if n < 5: continue
        print(key, n, {k: round(sum(v.values()) / len(v), 1) for k, v in c.items()})
PY
cat "$OUT"/unsplit256.txt "$OUT"/split.txt "$OUT"/ycb_unsplit256.txt "$OUT"/s512.txt; tail -3 "$OUT"/unsplit256.log
