#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: per kernel-dispatch counter sums, in dispatch order.
usage: python tools/pmc_summary.py gpurun_out/pmc_TAG [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "fft_pass"
rows = OrderedDict()
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if filt not in r["Kernel_Name"]:
            continue
        key = int(r["Dispatch_Id"])
        d = rows.setdefault(key, {"kernel": r["Kernel_Name"][:60], "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size"),
                                  "lds": r.get("LDS_Block_Size"), "vgpr": r.get("VGPR_Count")})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = []
for d in rows.values():
    for k in d:
        if k not in ("kernel", "grid", "wg", "lds", "vgpr") and k not in names:
            names.append(k)
for key, d in rows.items():
    print(f"dispatch {key}: grid={d['grid']} wg={d['wg']} lds={d['lds']} vgpr={d['vgpr']}")
    for n in names:
        if n in d:
            print(f"    {n:28s} {d[n]:.4g}")
# kernel stats
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read())
