#!/usr/bin/env python3
"""HBM traffic per launch of the axis-pass kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), with the guide's gfx950 correction: FETCH_SIZE
tallies 128-byte read requests at 64 B, so it is doubled; both counters are in KiB.
usage: python tools/pmc_traffic.py gpurun_out/pmct_TAG ALGORITHMIC_BYTES "workload text" > profiles/TAG_pmc_traffic.json
(the directory is what tools/pmc_traffic.sh writes)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, algorithmic, workload = sys.argv[1], int(sys.argv[2]), sys.argv[3]
per = defaultdict(lambda: defaultdict(float))     # counter -> dispatch -> sum over instances
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "fft_pass_kernel" not in r["Kernel_Name"]:
            continue
        per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
fetch, write = per.get("FETCH_SIZE", {}), per.get("WRITE_SIZE", {})
# dispatch ids differ between the two runs only by a constant; pair launches by order
rd = [fetch[k] * 1024 * 2 for k in sorted(fetch)]
wr = [write[k] * 1024 for k in sorted(write)]
n = min(len(rd), len(wr))
tot = [rd[i] + wr[i] for i in range(n)]
out = {
    "source": "tools/pmc_traffic.sh (separate --pmc passes: FETCH_SIZE, WRITE_SIZE), kernels of this round",
    "correction": "FETCH_SIZE*1024*2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
    "kernel": "dfft::fft_pass_kernel<...> (every instantiation the plan launches)",
    "workload": workload,
    "hbm_bytes_per_launch": sum(tot) / n if n else None,
    "min": min(tot) if n else None,
    "max": max(tot) if n else None,
    "read_bytes_per_launch": sum(rd[:n]) / n if n else None,
    "write_bytes_per_launch": sum(wr[:n]) / n if n else None,
    "algorithmic_bytes_per_launch": algorithmic,
    "dispatches": n,
}
print(json.dumps(out, indent=1))
