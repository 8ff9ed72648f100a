"""Fills BASELINE.md §3 (the target table) from the round's committed bench lines, profiles/<round>_bench_*.json -- by
script, so that the table is what the files say (the table stayed "—" for four rounds while the numbers
sat in DESIGN.md §6).

  python tools/fill_baseline_table.py r05"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line_of(path):
    try:
        txt = open(path).read().strip().splitlines()
        return json.loads(txt[-1]) if txt else None
    except Exception:  # noqa: BLE001
        return None


def fmt_rate(v):
    return "{:,.0f}".format(v).replace(",", " ") if v is not None else "—"


def parity_of(d):
    p = d.get("parity") or {}
    if not p:
        return "—"
    if p.get("bit_identical"):
        return "0 / 0 / 0 (bit-identical, %s objects, %s frames)" % (p.get("n"), p.get("frames"))
    return "%.2g rad / %.2g m / %.2g m" % (p.get("rot_max", 0), p.get("trans_max", 0), p.get("add_s_max", 0))


def row(label, d, note=""):
    if d is None:
        return "| %s | — | — | — | — | — |" % label
    cpu = d.get("cpu_baseline") or {}
    allc = cpu.get("all_cores") or {}
    roof = d.get("roofline") or {}
    frac = roof.get("frac")
    return "| %s | %s (%s ms / step%s) | %s | %s | %s | %s |" % (
        label, fmt_rate(d.get("value")), d.get("ms_per_step"), note,
        ("%.4f (kernel %s, %s ms)" % (frac, (roof.get("kernel") or "").split(" ")[0], roof.get("kernel_ms"))) if frac is not None else "n/a",
        parity_of(d), fmt_rate(cpu.get("value")) + (" (restatement, 1 thread)" if cpu.get("value") else ""),
        (fmt_rate(allc.get("value")) + " (N=%s)" % allc.get("cores")) if allc.get("value") else "—")


def main(tag):
    prof = os.path.join(ROOT, "profiles")
    b = {c: line_of(os.path.join(prof, "%s_bench_%s.json" % (tag, c))) for c in ("rbot64", "ycb21", "synth512", "chain8", "rbot4096")}
    rows = []
    r64 = b["rbot64"]
    cpu1 = (r64 or {}).get("cpu_baseline") or {}
    rows.append("| 1. single RBOT-style object, CPU path | %s (oracle restatement, 1 thread; `-march=native` build: %s) | n/a | n/a | %s | — |" % (
        fmt_rate(cpu1.get("value")), fmt_rate((cpu1.get("native_build") or {}).get("value")), fmt_rate(cpu1.get("value"))))
    rows.append(row("2. 64 objects, Region only, 200 lines × 7 it, 1×MI355X", r64))
    if b["rbot4096"]:
        rows.append(row("2b. the same, 4096 objects in one launch (batch sweep), 1×MI355X", b["rbot4096"]))
    rows.append(row("3. YCB-style scene, 21 objects, Region+Depth, 1×MI355X", b["ycb21"]))
    rows.append(row("4. 512 synthetic objects, Region+Depth — on **1**×MI355X (no 8-GPU node was available: see `projected_scaling` in the same file)", b["synth512"]))
    rows.append(row("5. 8-body 13-dof kinematic chain — one launch per frame on **1**×MI355X (the all-reduce path: `distributed_path_world1` in the same file)", b["chain8"]))
    header = ("| Config (BASELINE.json) | pose-updates/s | fraction of HBM roofline¹ | max pose error vs CPU restatement² | CPU 1-thread | CPU all-cores (N) |\n"
              "|---|---|---|---|---|---|\n")
    table = header + "\n".join(rows) + "\n"
    path = os.path.join(ROOT, "BASELINE.md")
    s = open(path).read()
    m = re.search(r"(## 3\. Target table[^\n]*\n\n)(\|.*?\n)(\n¹)", s, flags=re.S)
    assert m, "BASELINE.md §3 not found"
    s = s[:m.start(1)] + "## 3. Target table (filled by `tools/fill_baseline_table.py %s` from `profiles/%s_bench_*.json`)\n\n" % (tag, tag) + table + s[m.start(3):]
    open(path, "w").write(s)
    print(table)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r05")
