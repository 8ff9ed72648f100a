#!/bin/bash
# developer helper: SQ counter passes of tools/quick_bench.py for one library / batch / environment
#   tools/pmc_quick.sh <lib> <objects> [ENV=V ...]   (extra quick_bench flags through QB_FLAGS)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$(realpath $1); N=$2; shift 2
for kv in "$@"; do export "$kv"; done
TAG=$(basename $(dirname $LIB))_${N}_$(echo "$@" | tr ' =' '__')
OUT=$REPO/gpurun_out/pmcq/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/quick_bench.py --objects $N --steps 12 --warmup 2 $QB_FLAGS $LIB"
declare -A PASS
PASS[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
PASS[sq2]="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
PASS[ta]="TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
PASS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
for p in ${PASSES:-sq1 sq2}; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} --output-format csv -d "$OUT/$p" -- $CMD > "$OUT/$p.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
per = {}
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1]
        key = (name, int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1), int(row["Workgroup_Size"]),
               row.get("VGPR_Count", ""), row.get("LDS_Block_Size", ""), row.get("Scratch_Size", ""))
        d = per.setdefault(key, {}).setdefault(row["Counter_Name"], {})
        d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
for key, counters in sorted(per.items()):
    n = max(len(v) for v in counters.values())
    if n < 8:
        continue
    m = {c: sum(v.values()) / len(v) for c, v in counters.items()}
    print("kernel %s grid %d x %d vgpr %s lds %s scratch %s launches %d" % (key + (n,)))
    g = m.get
    if g("SQ_WAVE_CYCLES"):
        print("  parked %.3f issue-stall %.3f issuing %.3f" % (g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES"),
              g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")))
    gui = g("GRBM_GUI_ACTIVE", 0) / 8.0
    if gui:
        print("  kernel cycles (GRBM/8) %.0f" % gui)
    if g("SQ_WAVES"):
        print("  VALU/wave %.0f  SALU/wave %.0f  LDS/wave %.0f  VMEM/wave %.0f  per workgroup VALU %.0f" % (
            g("SQ_INSTS_VALU", 0) / g("SQ_WAVES"), g("SQ_INSTS_SALU", 0) / g("SQ_WAVES") if g("SQ_INSTS_SALU") else -1,
            g("SQ_INSTS_LDS", 0) / g("SQ_WAVES") if g("SQ_INSTS_LDS") else -1,
            g("SQ_INSTS_VMEM", 0) / g("SQ_WAVES") if g("SQ_INSTS_VMEM") else -1,
            g("SQ_INSTS_VALU", 0) / key[1]))
    if g("SQ_BUSY_CYCLES") and g("SQ_WAVE_CYCLES"):
        print("  SQ_BUSY_CYCLES %.0f WAVE_CYCLES %.0f ACTIVE_INST_VALU %.0f" % (g("SQ_BUSY_CYCLES"), g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU", 0)))
    for name in ("TA_TA_BUSY_sum", "TA_BUSY_avr", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum",
                 "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum",
                 "FETCH_SIZE", "WRITE_SIZE"):
        if g(name) is not None:
            print("  %s %.0f" % (name, g(name)))
    if g("SQ_LDS_IDX_ACTIVE"):
        print("  LDS bank conflict frac %.3f" % (g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")))
PY
