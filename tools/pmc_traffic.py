#!/usr/bin/env python3
"""HBM traffic per launch of the axis-pass kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), with the guide's gfx950 correction: FETCH_SIZE
tallies 128-byte read requests at 64 B, so it is doubled; both counters are in KiB.
usage: python tools/pmc_traffic.py gpurun_out/pmct_TAG ALGORITHMIC_BYTES "workload text" [KERNEL_SUBSTRING] > profiles/TAG_pmc_traffic.json
(the directory is what tools/pmc_traffic.sh writes; KERNEL_SUBSTRING defaults to fft_pass_kernel, "dfft::fft_" takes the real and
Bluestein kernels in as well; per_kernel lists every instantiation on its own)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, algorithmic, workload = sys.argv[1], int(sys.argv[2]), sys.argv[3]
want = sys.argv[4] if len(sys.argv) > 4 else "fft_pass_kernel"
per = defaultdict(lambda: defaultdict(float))     # counter -> dispatch -> sum over instances
names = {}                                        # counter -> dispatch -> kernel name
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        names.setdefault(r["Counter_Name"], {})[int(r["Dispatch_Id"])] = r["Kernel_Name"]
fetch, write = per.get("FETCH_SIZE", {}), per.get("WRITE_SIZE", {})
# dispatch ids differ between the two runs only by a constant; pair launches by order
rd = [fetch[k] * 1024 * 2 for k in sorted(fetch)]
wr = [write[k] * 1024 for k in sorted(write)]
n = min(len(rd), len(wr))
tot = [rd[i] + wr[i] for i in range(n)]
by_kernel = defaultdict(lambda: [0.0, 0.0, 0])
fk = sorted(fetch)
for i in range(n):
    k = names.get("FETCH_SIZE", {}).get(fk[i], "?")
    k = k.replace("void dfft::", "").split("(")[0]
    by_kernel[k][0] += rd[i]; by_kernel[k][1] += wr[i]; by_kernel[k][2] += 1
def library_sha256():
    """sha256 of the libdfft_amd.so next to this tree: bench.py only carries a traffic figure measured on the library it loads"""
    import hashlib
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "distributedfft_amd", "libdfft_amd.so")
    try:
        return hashlib.sha256(open(so, "rb").read()).hexdigest()
    except OSError:
        return None


out = {
    "library_sha256": library_sha256(),
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
    "per_kernel": {k: {"launches": v[2], "read_bytes_per_launch": v[0] / v[2], "write_bytes_per_launch": v[1] / v[2]} for k, v in by_kernel.items()},
}
print(json.dumps(out, indent=1))
