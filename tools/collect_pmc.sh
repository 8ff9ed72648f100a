#!/bin/bash
# PMC passes of the default bench command (MI355X_MICROARCH.md: 8 SQ slots, 4 TCC slots per pass; FETCH_SIZE and
# WRITE_SIZE in passes of their own; --pmc never together with the hip/hsa trace domains).  Writes
# gpurun_out/pmc/<pass>/ and the summary gpurun_out/pmc_summary.json, committed as profiles/rNN_pmc_<config>.json.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CONFIG=${1:-rbot64}
OUT=$REPO/gpurun_out/pmc_$CONFIG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --config $CONFIG --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeats 1"
declare -A PASS
PASS[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
PASS[sq2]="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
PASS[ta]="TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
PASS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
for p in sq1 sq2 ta tcc fetch write; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} --output-format csv -d "$OUT/$p" -- $CMD > "$OUT/$p.log" 2>&1
done
python "$REPO/tools/pmc_summary2.py" "$OUT" "$REPO/gpurun_out/pmc_summary_$CONFIG.json" "$CONFIG"
