"""Developer tool: which source lines the scratch (spill) instructions of a kernel belong to.
  hipcc ... -gline-tables-only --cuda-device-only -S -o dev.s m3t_hip_api.hip
  python tools/scratch_lines.py dev.s tracking_step_tree_kernel
Counts scratch_load / scratch_store per (file, line) from the .loc directives in front of them; also prints the
kernel's instruction count per source function range when --hist is given."""
import collections
import re
import sys

path, kernel = sys.argv[1], sys.argv[2]
files = {}
counts = collections.Counter()
total = collections.Counter()
inside = False
loc = None
for line in open(path, errors="replace"):
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if line.startswith(kernel + ":"):
        inside = True
        continue
    if inside and line.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = line.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    total[loc] += 1
    if t.startswith("scratch_"):
        counts[(loc, t.split()[0])] += 1
n = sum(counts.values())
print("%s: %d instructions, %d scratch instructions" % (kernel, sum(total.values()), n))
for (loc, op), c in sorted(counts.items(), key=lambda kv: -kv[1])[:40]:
    print("%5d  %-22s %s:%s" % (c, op, loc[0], loc[1]))
if "--hist" in sys.argv:
    per_file = collections.Counter()
    for loc, c in total.items():
        per_file[(loc[0], loc[1] // 50 * 50)] += c
    for (f, l), c in sorted(per_file.items()):
        print("%6d  %s:%d-%d" % (c, f, l, l + 49))
