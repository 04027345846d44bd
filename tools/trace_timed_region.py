#!/usr/bin/env python3
"""Per-kernel statistics of the LAST n launches of the axis-pass kernels in a rocprofv3 kernel trace.
usage: python tools/trace_timed_region.py <..._kernel_trace.csv> N [substring]
`bench.py` tunes the placement of its buffers and warms up before the timed steps, all in one process; the --stats summary of that
process therefore averages the tuning trials (on buffers that were then discarded) and the warm-up in.  The timed region is the last
steps * 6 launches of the trace: this prints their statistics, which is what roofline.avg_launch_ms of the bench line must agree with."""
import csv
import sys
from collections import OrderedDict

path, n = sys.argv[1], int(sys.argv[2])
want = sys.argv[3] if len(sys.argv) > 3 else "fft_pass_kernel"
rows = [r for r in csv.DictReader(open(path)) if want in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
per = OrderedDict()
for r in last:
    per.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
allms = [v for d in per.values() for v in d]
print(f"{path}: {len(rows)} launches of *{want}* in the process, statistics of the last {len(last)} (the timed region)")
print(f"{'calls':>6s} {'avg ms':>9s} {'min ms':>9s} {'max ms':>9s}  kernel")
for k, d in per.items():
    print(f"{len(d):6d} {sum(d) / len(d):9.4f} {min(d):9.4f} {max(d):9.4f}  {k}")
print(f"{len(allms):6d} {sum(allms) / len(allms):9.4f} {min(allms):9.4f} {max(allms):9.4f}  all")
whole = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
print(f"{len(whole):6d} {sum(whole) / len(whole):9.4f} {min(whole):9.4f} {max(whole):9.4f}  every launch of the process (tuning trials and warm-up included: what --stats averages)")
