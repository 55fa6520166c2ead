#!/usr/bin/env python3
"""The per-round table of DESIGN.md section 4 AND the round's section of profiles/README.md from the bench lines under profiles/
(so that no number is typed by hand, and the README cannot lag a round behind again: VERDICT r05 item 7).
    python tools/design_table.py r06"""
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [("bench", "venice-1778 (headline, default command)"), ("bench_driver_flags", "venice-1778, the driver's flags (`--steps 20 --warmup 5`)"),
        ("bench_forced_e0_lpl", "venice-1778, `e0_lpl` forced (`POVAR_E0_CK=0`)"), ("bench_huber", "venice-1778, HUBER"),
        ("bench_local", "venice-1778, `--popularity local`"), ("bench_zipf05", "venice-1778, Zipf(0.5)"),
        ("bench_uniform", "venice-1778, uniform popularity"), ("bench_step2", "venice-1778, step 2"),
        ("bench_step2_forced_e0_lpl_h", "venice-1778, step 2, `e0_lpl_h` forced"), ("bench_step2_huber", "venice-1778, step 2, HUBER"),
        ("bench_final_huber", "final-13682, HUBER"), ("bench_trafalgar", "trafalgar-257"),
        ("bench_trafalgar_per_term_kernels", "trafalgar-257, per-term kernels forced (`POVAR_RES=0`)"), ("bench_ladybug", "ladybug-49"),
        ("bench_ladybug_per_term_kernels", "ladybug-49, per-term kernels forced"), ("bench_deterministic", "venice-1778, `POVAR_DETERMINISTIC=1`"), ("bench_deterministic_huber", "venice-1778, HUBER, `POVAR_DETERMINISTIC=1`"),
        ("bench_deterministic_gather", "venice-1778, `POVAR_DETERMINISTIC=1 POVAR_DET_CK=0` (gather form)"),
        ("bench_deterministic_step2", "venice-1778, step 2, `POVAR_DETERMINISTIC=1`"),
        ("bench_deterministic_step2_gather", "venice-1778, step 2, `POVAR_DETERMINISTIC=1 POVAR_DET_CK=0` (gather form)")]
lines = []
lines.append("| workload | terms/s | term kernel(s) (the library's timing of the two pairs) | pair time (events) | bytes per E0 (measured / every array once) | fraction (measured / once) |")
lines.append("|---|---|---|---|---|---|")
for f, name in ROWS:
    path = os.path.join(ROOT, "profiles", f"{TAG}_{f}.json")
    if not os.path.exists(path):
        continue
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r, el = d["roofline"], d["config"]["e0_layout"]
    res = el.get("series_kernel", "").startswith("resident")
    step2 = "step 2" in name
    if res:
        kern = f"`series_res` ({el['series_tune_us_per_term']['per_term_kernels']:.1f} / {el['series_tune_us_per_term']['resident']:.1f} µs per term timed)"
    elif step2:
        kern = ("`e0_ck_h_det`" if el["e0_kernel_step2"] == 2 else "`lm_regular<OpE0H>` + `cm_scatter`" if "gather" in name else "`e0_ck_h`" if el["e0_kernel_step2"] else "`e0_lpl_h`") + (f" (timed {el['e0_tune_us_step2']['e0_lpl_h']:.1f} / {el['e0_tune_us_step2']['e0_ck_h']:.1f} µs)" if el["e0_tune_us_step2"]["e0_ck_h"] > 0 else "")
    else:
        kern = ("`e0_ck_det`" if el["e0_kernel"] == 7 else "`e0_lm_cached<false>` + `cm_scatter`" if "gather" in name else "`e0_ck`" if el["e0_kernel"] else "`e0_lpl`" if el["term_kernels"] == "lane per landmark" else "`e0_lm_cached`") + \
            (f" (timed {el['e0_tune_us']['e0_lpl']:.1f} / {el['e0_tune_us']['e0_ck']:.1f} µs)" if el["e0_tune_us"]["e0_ck"] > 0 else "")
    tr = r.get("traffic")
    once = r["once_bytes_per_launch"]
    ms = d["kernel_ms"]["e0"]
    frac_m = f"{tr / (ms * 1e-3) / 8e12:.2f}" if tr else "—"
    lines.append(f"| {name} | {d['value']:,.0f} | {kern} | {1e3 * ms:.1f} µs | {tr / 1e6:.0f} MB / {once / 1e6:.0f} MB | {frac_m} / {r['once_frac']:.2f} |" if tr else
                 f"| {name} | {d['value']:,.0f} | {kern} | {1e3 * ms:.1f} µs | — / {once / 1e6:.1f} MB | — / {r['once_frac']:.2f} |")

print("\n".join(lines))
dpath = os.path.join(ROOT, "DESIGN.md")
s = open(dpath).read()
B, E = f"<!-- {TAG.upper()}-TABLE-BEGIN -->", f"<!-- {TAG.upper()}-TABLE-END -->"
if B in s:
    a, b = s.index(B), s.index(E)
    open(dpath, "w").write(s[:a] + B + "\n" + "\n".join(lines) + "\n" + s[b:])
# the same rows as the round's section of profiles/README.md (between its markers; the hand-written file list stays below them)
rpath = os.path.join(ROOT, "profiles", "README.md")
r = open(rpath).read()
if B in r:
    a, b = r.index(B), r.index(E)
    open(rpath, "w").write(r[:a] + B + "\n" + "\n".join(lines) + "\n" + r[b:])
