"""Measuring tool: host API calls and kernels of the LAST call in a rocprofv3 --hip-trace --kernel-trace csv pair (one timeline).
usage: api_timeline.py <dir with *_hip_api_trace.csv and *_kernel_trace.csv> [name of the call's last kernel]"""
import csv, glob, sys
d = sys.argv[1]
last_kernel = sys.argv[2] if len(sys.argv) > 2 else "k_banded"
api = list(csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])))
ker = [r for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])) if "swh::" in r["Kernel_Name"]]
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(ker) if last_kernel in r["Kernel_Name"]]
lo = int(ker[ends[-2]]["End_Timestamp"]) if len(ends) > 1 else 0
hi = int(ker[ends[-1]]["End_Timestamp"]) + 60000
rows = [("K " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ker if lo < int(r["Start_Timestamp"]) < hi]
rows += [("A " + r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in api if lo < int(r["Start_Timestamp"]) < hi]
rows.sort(key=lambda x: x[1])
t0 = rows[0][1]
for name, s, e in rows:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {name}")
