#!/usr/bin/env python3
"""Runs the resident series of a shape with the stamped diagnostic library (tools/variants/res_stamps.py) and prints where
the cycles of ONE term go: per phase the median / maximum over the workgroups (wavefront 0 and the last wavefront), and the
spread of the phase starts across workgroups (who waits for whom).  usage: res_stamps_report.py shape [world]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {1: "top of the term", 2: "norms done: gather z", 3: "z gathered (before B1)", 4: "behind B1: forward", 5: "forward done (before B2)",
         6: "g = G u done (before B3)", 7: "backward done (before B4)", 8: "records written (before B5)",
         9: "owner's records gathered (before B6)", 10: "z published (before B7)", 12: "loop exit"}


def main():
    out = "/tmp/res_stamps.txt"
    env = dict(os.environ, POVAR_LIB=os.path.join(ROOT, "build", "libpovar_hip_res_stamps.so"), POVAR_RES_STAMPS_OUT=out, POVAR_RES="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "res_term_time.py")] + sys.argv[1:], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:])
    import numpy as np
    raw = np.loadtxt(out, dtype=np.uint64).reshape(2, -1, 2, 16).astype(np.float64)
    a, rt = raw[0], raw[1]  # shader-clock stamps (per XCD: only differences inside a workgroup mean anything), 100 MHz real time
    used = a[:, 0, 2] > 0
    a, rt = a[used], rt[used]
    print(f"{a.shape[0]} workgroups; cycles (s_memtime, 100 MHz reference clock ticks x ...: raw counter units)")
    t0 = a[:, :, 1].min()
    order = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
    for w, wn in ((0, "wavefront 0"), (1, "last wavefront")):
        print(f"-- {wn}")
        prev = None
        for n in order:
            v = a[:, w, n]
            if not (v > 0).all():
                continue
            rel = v - t0
            line = f"  {n:2d} {NAMES[n]:45s} at median {np.median(rel):9.0f}  min {rel.min():9.0f}  max {rel.max():9.0f}"
            if prev is not None:
                dlt = v - a[:, w, prev]
                line += f"   since {prev:2d}: median {np.median(dlt):8.0f} max {dlt.max():8.0f}"
            print(line)
            prev = n
    # the same phases on the chip-wide 100 MHz clock: when the FIRST and the LAST workgroup reach each stamp, from the moment the
    # first workgroup starts the term (10 ns resolution) -- who waits for whom across workgroups
    r0 = rt[:, :, 1].min()
    print("-- chip-wide (s_memrealtime), microseconds after the first workgroup's top of the term: first / median / last workgroup")
    for n in order:
        v = rt[:, :, n]
        if not (v > 0).all():
            continue
        x = (v.min(axis=1) - r0) / 100.0
        y = (v.max(axis=1) - r0) / 100.0
        print(f"  {n:2d} {NAMES[n]:45s} first wave {x.min():6.2f} {np.median(x):6.2f} {x.max():6.2f}   last wave {y.min():6.2f} {np.median(y):6.2f} {y.max():6.2f}")
    if (a[:, 0, 12] > 0).all():
        print(f"term length (stamp 1 of this term -> 1 of the next is not stamped; kernel entry -> exit): {np.median(a[:, 0, 12] - a[:, 0, 0]):.0f} ticks for the launch")


if __name__ == "__main__":
    main()
