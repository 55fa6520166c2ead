#!/usr/bin/env python3
"""Which device functions of a translation unit changed between two builds?  Compares the gfx950 assembly of every kernel
(labels renumbered, comments dropped) instruction by instruction -- the check behind "the PMC passes of the other kernels
still hold" when one instantiation is edited (profiles/r06_isa_unchanged.txt).

usage: hipcc --offload-arch=gfx950 <CXXFLAGS> --cuda-device-only -S -o old.s unit.hip   (on both trees), then
       isa_same.py old.s new.s"""
import re
import subprocess
import sys


def funcs(path):
    out, cur, buf = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z[^:]*):", line)
        if m and cur is None:
            cur, buf = m.group(1), []
        elif cur and line.startswith(".Lfunc_end"):
            out[cur], cur = buf, None
        elif cur:
            buf.append(line)
    return out


def body(buf):
    lines = [re.sub(r";.*", "", x).strip() for x in buf if not x.startswith(".L") and not x.strip().startswith(";")]
    return [re.sub(r"\.LBB\d+_\d+", "L", x) for x in lines if x]


def main():
    a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
    names = sorted(set(a) | set(b))
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    same, diff = 0, []
    for n, d in zip(names, dem):
        sa, sb = body(a.get(n, [])), body(b.get(n, []))
        if sa == sb:
            same += 1
        else:
            diff.append((d.split("(")[0], len(sa), len(sb)))
    print(f"{same} of {len(names)} device functions identical instruction by instruction")
    for d, la, lb in diff:
        print(f"  differs: {d}   {la} -> {lb} instructions")


if __name__ == "__main__":
    main()
