"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE against known byte counts (tools/ubench_counters.hip).

  python tools/counter_calibration.py <dir with fetch/ and write/ rocprofv3 outputs> <ubench stdout> [out.txt]

Prints, per access pattern, what the counter reported (KB x 1024) divided by the bytes the kernel asked for and by
the bytes at a 64-byte and a 128-byte fill granule; tools/pmc_summary3.py reads the factors back from the text file
(lines 'factor <pattern> <counter> <reported / bytes moved>')."""
import csv
import glob
import os
import sys


def reported(folder, counter):
    per = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            per[name] = per.get(name, 0.0) + float(row["Counter_Value"])
    return per


def main(out_dir, known_path, target=None):
    known = {}
    for line in open(known_path):
        p = line.split()
        if len(p) == 4 and p[0].startswith("cal_"):
            known[p[0]] = [int(x) for x in p[1:]]
    fetch = reported(os.path.join(out_dir, "fetch"), "FETCH_SIZE")
    write = reported(os.path.join(out_dir, "write"), "WRITE_SIZE")
    lines = ["rocprofv3 FETCH_SIZE / WRITE_SIZE (KB x 1024) against known byte counts, tools/ubench_counters.hip, one launch each",
             "%-24s %-10s %14s %14s %10s %10s %10s" % ("pattern", "counter", "reported B", "asked-for B", "/asked", "/64B gran", "/128B gran")]
    factors = []
    for name, (asked, g64, g128) in known.items():
        for counter, table in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            if name not in table:
                continue
            is_write = name.startswith("cal_write")
            if (counter == "WRITE_SIZE") != is_write:
                # the other side of a pattern: reads of a write kernel (read-for-ownership?) and the reverse, as a note
                if table[name] * 1024 > 0.01 * asked:
                    lines.append("%-24s %-10s %14.0f %14s   (the pattern's other direction)" % (name, counter, table[name] * 1024, "-"))
                continue
            rep = table[name] * 1024
            lines.append("%-24s %-10s %14.0f %14d %10.3f %10.3f %10.3f" % (name, counter, rep, asked, rep / asked, rep / g64, rep / g128))
            factors.append("factor %s %s %.4f" % (name, counter, rep / g64))
    text = "\n".join(lines + [""] + factors) + "\n"
    sys.stdout.write(text)
    if target:
        open(target, "w").write(text)


if __name__ == "__main__":
    main(*sys.argv[1:4])
