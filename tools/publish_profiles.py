"""Copies the round's evidence from gpurun_out/ (scratch) to profiles/ (tracked): bench lines, rocprofv3 kernel
statistics, PMC summaries, HBM traffic blocks, phase timings, the GPU test log.  Later collection runs of the same
round (r03f after r03: the configurations whose kernel changed once more) take precedence; a PMC summary that exists
in both is merged counter by counter (the later run may hold passes the first lacked) and its ratios are recomputed.

  python tools/publish_profiles.py r03 r03f
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary3 import ratios_of  # noqa: E402


def main(rounds):
    tag = rounds[0]
    dst = os.path.join(ROOT, "profiles")
    seen = {}
    for r in rounds:
        src = os.path.join(ROOT, "gpurun_out", r)
        for name in sorted(os.listdir(src)):
            if name.endswith(".err") or name.startswith("stats_") or name == "collect.log":
                continue
            path = os.path.join(src, name)
            if os.path.isdir(path):
                continue
            out = os.path.join(dst, "%s_%s" % (tag, name))
            if name.startswith("pmc_") and name in seen:
                a, b = json.load(open(out)), json.load(open(path))
                for k, v in b["kernels"].items():
                    if k in a["kernels"]:
                        a["kernels"][k]["per_launch_means"].update(v["per_launch_means"])
                    else:
                        a["kernels"][k] = v
                for k in a["kernels"].values():
                    k["ratios"] = ratios_of(k["per_launch_means"])
                a["command"] = a["command"] + " ; " + b["command"]
                json.dump(a, open(out, "w"), indent=1)
            elif name.startswith("hbm_traffic_") and name in seen and not json.load(open(path))["kernels"]:
                pass  # a later run without the size passes does not replace the traffic block
            else:
                shutil.copyfile(path, out)
            seen[name] = r
    for name, r in sorted(seen.items()):
        print("%-32s from gpurun_out/%s" % (name, r))


if __name__ == "__main__":
    main(sys.argv[1:] or ["r04"])
