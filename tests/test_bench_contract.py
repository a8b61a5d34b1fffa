"""bench.py without a GPU: the roofline arithmetic and the link between the workloads it times and the committed
rocprof summaries it replays its executed-instruction fields from."""
import ast
import json
import os
import re

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def default_tags():
    """Tags of the workloads the default run times, read from the Work2D / Work3D calls in bench.py's source."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tags = ["complex_world_1920x1080_s64_l2"]  # the headline: the argparse defaults
    for m in re.finditer(r'Work2D\(rt, torch, ctx, "(\w+)", (\d+), (\d+), (\d+), (\d+), \d+\)', src):
        name, w, h, spp, ls = m.groups()
        tags.append(f"{name}_{w}x{h}_s{spp}_l{ls}")
    for m in re.finditer(r"Work3D\(rt, torch, ctx, (\d+), (\d+), (\d+), \d+, (True|False)\)", src):
        w, h, spp, bvh = m.groups()
        tags.append(f"3d_{'bvh' if bvh == 'True' else 'linear'}_{w}x{h}_s{spp}")
    return tags


def test_argparse_defaults_are_the_headline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            flag = node.args[0].value
            for kw in node.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[flag] = kw.value.value
    assert (defaults["--scene"], defaults["--width"], defaults["--height"]) == ("complex_world", 1920, 1080)
    assert (defaults["--spp"], defaults["--lambda-samples"], defaults["--depth"]) == (64, 2, 10)
    assert defaults["--gpus"] == 1 and defaults["--mode"] == "weak"


def test_every_default_workload_has_committed_counters():
    tags = default_tags()
    assert len(tags) == 7 and len(set(tags)) == 7
    for tag in tags:
        prof = bench.profile_counters(tag)
        assert prof is not None, f"no profiles/r*/pmc_{tag}.json"
        assert os.path.exists(os.path.join(ROOT, prof["source"]))
        assert 0.3 < prof["valu_issue_frac"] < 1.2
        assert prof["hbm_bytes_per_launch"] > 0 and prof["valu_insts_per_launch"] > 0
        assert prof["flop_f32_per_launch_full_waves"] > 0 and prof["flop_f64_per_launch_full_waves"] > 0
        stats = [f for f in os.listdir(os.path.dirname(os.path.join(ROOT, prof["source"]))) if f == f"kernel_stats_{tag}.csv"]
        assert stats, f"no kernel_stats_{tag}.csv beside {prof['source']}"
    assert bench.profile_counters("no_such_workload_1x1_s1_l1") is None


def fractions(obj, path="roofline"):
    """Every `frac`-like number in a roofline object, with its path."""
    for k, v in obj.items():
        if isinstance(v, dict):
            yield from fractions(v, f"{path}.{k}")
        elif "frac" in k and isinstance(v, (int, float)):
            yield f"{path}.{k}", v


def test_roofline_arithmetic():
    st = {"launches": 5, "kernel_ms": 100.0, "rays": 5 * 575e6, "searches": 5 * 440e6, "bvh_node_visits": 5 * 3.1e9,
          "sphere_tests": 5 * 1.7e9}
    r = bench.roofline("2d", st, "render2d_kernel", 200, 2.0e9, "complex_world_1920x1080_s64_l2")
    flop = 575e6 * (15 * 200 + 60)
    assert abs(r["achieved"] - flop / 20e-3 / 1e12) < 1e-9
    assert r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-12 and r["frac"] < 1.0
    assert r["kernel_ms"] == 20.0 and r["bound"] == "valu" and r["unit"] == "TFLOP/s"
    assert abs(r["frac_searches_executed"] - (440e6 * 15 * 200 + 575e6 * 60) / 20e-3 / 1e12 / 157.3) < 1e-12
    assert r["frac_searches_executed"] < r["frac"]
    assert r["counters_from"].startswith("profiles/") and os.path.exists(os.path.join(ROOT, r["counters_from"]))
    # one byte model for HBM: the counters' bytes, the same number `traffic` carries
    assert abs(r["hbm"]["achieved"] - r["traffic"] / 20e-3 / 1e9) < 1e-6 and r["hbm"]["model"].startswith("PMC")
    # executed FLOP by type, from the hardware's own count, each over the peak of its type
    assert r["executed_f32"]["peak"] == 157.3 and r["executed_f64"]["peak"] == 78.6
    for ex in (r["executed_f32"], r["executed_f64"]):
        assert 0.0 < ex["frac"] < 1.0 and abs(ex["frac"] - ex["achieved"] / ex["peak"]) < 1e-12
    assert 0.2 < r["valu_active_lane_share"] <= 1.0
    r3 = bench.roofline("3d_linear", st, "render3d_kernel<false>", 218, 1.0e9, "3d_linear_1920x1080_s128")
    assert abs(r3["achieved"] - 575e6 * (23 * 218 + 60) / 20e-3 / 1e12) < 1e-9 and "frac_searches_executed" not in r3
    rb = bench.roofline("3d_bvh", st, "render3d_kernel<true>", 218, 1.0e9, "3d_bvh_1920x1080_s128")
    assert abs(rb["achieved"] - (3.1e9 * 2 * 30 + 1.7e9 * 23 + 575e6 * 60) / 20e-3 / 1e12) < 1e-9
    assert rb["frac"] < 1.0 and abs(rb["reference_scan_equivalent"]["achieved"] - r3["achieved"]) < 1e-9
    # a workload nobody profiled still gets its measured fields, and says what its HBM figure is
    rn = bench.roofline("2d", st, "render2d_kernel", 11, 1.0e9, "some_world_10x10_s1_l1")
    assert rn["traffic"] is None and "counters_from" not in rn and rn["hbm"]["model"].startswith("stand-in")
    json.dumps([r, r3, rb, rn])


def test_no_fraction_of_a_committed_profile_exceeds_its_peak():
    """With each workload's own launch time (rocprof's average in the kernel_stats file beside the counters), every
    fraction in the roofline object stays below 1 - the issue share of the vector pipes aside, whose counter sums
    overlapping per-wave execution cycles and reads up to about 1.15 on a saturated pipe."""
    import csv
    for tag in default_tags():
        prof = bench.profile_counters(tag)
        stats = os.path.join(ROOT, os.path.dirname(prof["source"]), f"kernel_stats_{tag}.csv")
        row = max(csv.DictReader(open(stats)), key=lambda r: float(r["TotalDurationNs"]))
        assert "render" in row["Name"]
        ms = float(row["AverageNs"]) / 1e6
        kind = "3d_bvh" if "3d_bvh" in tag else "3d_linear" if "3d_linear" in tag else "2d"
        # the model's inputs do not matter here: only the counter-derived fields are checked
        st = {"launches": 1, "kernel_ms": ms, "rays": 1.0, "searches": 1.0, "bvh_node_visits": 1.0, "sphere_tests": 1.0}
        r = bench.roofline(kind, st, "k", 1, 1.0, tag)
        assert "executed_f32" in r and "executed_f64" in r, f"{tag}: no FLOP counters in {prof['source']}"
        for path, v in fractions(r):
            limit = 1.2 if path.endswith("valu_issue_frac") else 1.0
            assert 0.0 <= v < limit, f"{tag}: {path} = {v}"


def test_plain_multi_gpu_invocation_becomes_the_launcher(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE starts N child ranks (no torch, no HIP in the parent) and
    hands back the first failing rank's exit code.  Here the children are a stand-in script."""
    import subprocess
    import sys
    script = tmp_path / "fake_bench.py"
    script.write_text("import os, sys\n"
                      "r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "open(os.path.join(os.path.dirname(__file__), f'rank{r}'), 'w').write(' '.join(sys.argv[1:]))\n"
                      "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r and 7 or 0)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "2"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "FAIL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.spawn_ranks(3) == 0
    assert sorted(p.name for p in tmp_path.glob("rank*")) == ["rank0", "rank1", "rank2"]
    assert (tmp_path / "rank1").read_text() == "--gpus 3 --steps 2"
    monkeypatch.setenv("FAIL_RANK", "1")
    assert bench.spawn_ranks(3) == 7
    # and main() takes that road before importing torch
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("spawn_ranks(args.gpus)") < src.index("    import torch\n")
    assert "launch with torch.distributed.run" not in src
    del subprocess
