#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel + memory-copy trace: how much of the exchange (device-to-device
copies of the in-process transport) ran concurrently with axis-pass kernels."""
import csv, glob, os, sys
d = sys.argv[1]
def load(pattern, name_key, filt=None):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        for r in csv.DictReader(open(f)):
            nm = r.get(name_key, "")
            if filt and filt not in nm:
                continue
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
    return sorted(out)
kern = load("*kernel_trace.csv", "Kernel_Name", "fft_")
cop = load("*memory_copy_trace.csv", "Name")
blit = load("*kernel_trace.csv", "Kernel_Name", "copyBuffer")
moves = cop + blit
def union(iv):
    iv = sorted((a, b) for a, b, _ in iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out
def total(u): return sum(b - a for a, b in u)
def inter(u1, u2):
    i = j = 0; t = 0
    while i < len(u1) and j < len(u2):
        a = max(u1[i][0], u2[j][0]); b = min(u1[i][1], u2[j][1])
        if b > a: t += b - a
        if u1[i][1] < u2[j][1]: i += 1
        else: j += 1
    return t
uk, um = union(kern), union(moves)
print(f"axis-pass kernel launches: {len(kern)}, busy {total(uk)/1e6:.3f} ms")
print(f"exchange copies (memcpy records {len(cop)}, blit kernels {len(blit)}): busy {total(um)/1e6:.3f} ms")
if um:
    ov = inter(uk, um)
    print(f"copy time overlapped with axis-pass kernels: {ov/1e6:.3f} ms = {100.0*ov/total(um):.1f} % of the copy time")
    span = max(b for _, b in uk + um) - min(a for a, _ in uk + um)
    print(f"wall span of the traced work: {span/1e6:.3f} ms; sum of busy times {(total(uk)+total(um))/1e6:.3f} ms")
