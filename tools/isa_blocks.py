#!/usr/bin/env python3
"""Basic-block table of one kernel in a hipcc -S listing: label, vector / scalar / LDS / memory instruction counts and
the branch that ends the block.  Usage: python tools/isa_blocks.py listing.s mangled_kernel_name"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blk, rows = "entry", []
cnt = dict(v=0, s=0, ds=0, mem=0, f64=0, trans=0)
def flush(tail=""):
    global cnt
    rows.append((blk, dict(cnt), tail))
    cnt = dict(v=0, s=0, ds=0, mem=0, f64=0, trans=0)
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            flush()
            blk = t.split(":")[0]
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        flush()
        blk = m.group(1)
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cnt["v"] += 1
        if "_f64" in op:
            cnt["f64"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
            cnt["trans"] += 1
    elif op.startswith("ds_"):
        cnt["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cnt["mem"] += 1
    elif op.startswith("s_"):
        cnt["s"] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            flush(t)
            blk = blk + "'"
flush()
tot = dict(v=0, s=0, ds=0, mem=0, f64=0, trans=0)
for b, c, tail in rows:
    for k in tot:
        tot[k] += c[k]
    if sum(c.values()):
        print(f"{b:14s} v={c['v']:4d} (f64 {c['f64']:3d}, trans {c['trans']:2d}) s={c['s']:4d} ds={c['ds']:3d} mem={c['mem']:3d}  {tail}")
print("total", tot)
