#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel with
VGPRs / AGPRs / SGPRs / scratch / occupancy / static LDS.
usage: python tools/resource_usage.py remarks.txt [> profiles/rN_kernel_resources.txt]
(the remarks come from `make -C distributedfft_amd/csrc resources`)"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except Exception:  # noqa: BLE001
        return names


def main():
    txt = open(sys.argv[1]).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        name = b.split("\n")[0].split(" [")[0].strip()

        def g(key):
            m = re.search(re.escape(key) + r": (\S+)", b)
            return m.group(1) if m else "?"

        rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"),
                     g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
    names = demangle([r[0] for r in rows])
    print(f"{'kernel':118s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'occ':>4s} {'lds':>7s}")
    for n, r in zip(names, rows):
        n = n.replace("dfft::", "").replace("void ", "")
        n = re.sub(r"\(dfft::PassArgs\)|\(PassArgs\)", "", n)
        print(f"{n[:118]:118s} {r[1]:>5s} {r[2]:>5s} {r[3]:>5s} {r[4]:>8s} {r[5]:>4s} {r[6]:>7s}")


if __name__ == "__main__":
    main()
