"""Measuring tool: start / end of every kernel of the LAST calls in a rocprofv3 --kernel-trace csv, and the gaps between them.
usage: kernel_gaps.py <kernel_trace.csv> [kernels per call]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "swh::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
last = rows[-3 * per:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
for i, r in enumerate(last):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    print(f"{name:48s} stream {r.get('Stream_Id', '?'):>3s} start {s/1e3:9.1f} us  end {e/1e3:9.1f} us  length {(e-s)/1e3:8.1f}")
