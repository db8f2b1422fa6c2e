#!/usr/bin/env python3
"""Turns `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin or a file) into a table:
kernel, VGPRs, AGPRs, scratch bytes per lane, waves per SIMD, LDS bytes.  `make -C leg-kilo_amd/csrc resource-usage 2>&1 |
python tools/resource_table.py [filter ...]`."""
import re
import subprocess
import sys


def main():
    args = sys.argv[1:]
    src = sys.stdin
    if args and args[0].endswith(".txt"):
        src = open(args[0])
        args = args[1:]
    txt = src.read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].split(" [-R")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    seen = set()
    print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'waves':>5s} {'LDS':>6s}")
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*", "", d)
        if d in seen or (args and not any(a in d for a in args)):
            continue
        seen.add(d)
        vals = (g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
        print(f"{d[:78]:78s} {vals[0]:5d} {vals[1]:5d} {vals[2]:8d} {vals[3]:5d} {vals[4]:6d}")


if __name__ == "__main__":
    main()
