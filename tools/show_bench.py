#!/usr/bin/env python3
"""One line per configuration of a bench.py output line: python tools/show_bench.py <file with the JSON line>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])


def show(name, r, extra=""):
    f32 = r.get("executed_f32", {}).get("frac", float("nan"))
    f64 = r.get("executed_f64", {}).get("frac", float("nan"))
    print(f"{name:70s} {r['kernel_ms']:7.2f} ms  frac {r['frac']:.3f}  searched {r.get('frac_searches_executed', float('nan')):.3f}  "
          f"issue {r.get('valu_issue_frac', float('nan')):.3f}  f32 {f32:.3f}  f64 {f64:.3f}  hbm {r['hbm']['frac']:.4f}  {extra}")


show("headline: " + d["config"]["workload"][:58], d["roofline"], f"{d['value']:.0f} Mrays/s, {d['ms_per_step']:.2f} ms/step")
for c in d.get("configs", []):
    show(c["workload"][:70], c["roofline"], f"{c['Mrays_per_s']:.0f} Mrays/s")
if "cpu_baseline" in d:
    print(d["cpu_baseline"])
