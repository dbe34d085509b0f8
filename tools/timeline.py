"""Print the kernel timeline of a rocprofv3 --kernel-trace CSV (start, duration, queue), e.g. to see how the two
pipeline lanes of a scope overlap.  usage: python tools/timeline.py <kernel_trace.csv> [first] [count]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "swh::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    name = r["Kernel_Name"].split("(")[0].replace("void swh::", "").replace("swh::", "")[:30]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name:32s} q{r['Queue_Id']:>3s}  start {s / 1e3:9.1f} us  end {e / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f}")
