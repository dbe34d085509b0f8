"""Prints config, GCUPS per call, GCUPS over the kernels, dominant kernel, kernel ms and the parity flag of `bench.py --only-config` lines read from stdin."""
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print(d.get("config"), d.get("value"), d.get("gcups_kernels"), d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("kernel_ms"), d.get("parity_vs_oracle"))
