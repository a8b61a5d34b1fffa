#!/bin/bash
# Every workload bench.py times by default, through tools/profile_r3.sh (run on the GPU box from the repo root);
# afterwards, here: python tools/collect_profiles.py <tag> profiles/r03 <kernel> for each tag (tools/collect_all_r3.sh).
set -u
P="bash tools/profile_r3.sh"
$P complex_world_1920x1080_s64_l2 > gpurun_out/p1.log 2>&1 || { tail -5 gpurun_out/p1.log; exit 1; }
$P complex_world_3840x2160_s64_l2 --width 3840 --height 2160 > gpurun_out/p2.log 2>&1 || { tail -5 gpurun_out/p2.log; exit 1; }
$P cornell_box_3840x2160_s16_l16 --scene cornell_box --width 3840 --height 2160 --spp 16 --lambda-samples 16 > gpurun_out/p3.log 2>&1 || { tail -5 gpurun_out/p3.log; exit 1; }
$P cornell_box_absorption_3840x2160_s16_l16 --scene cornell_box_absorption --width 3840 --height 2160 --spp 16 --lambda-samples 16 > gpurun_out/p4.log 2>&1 || { tail -5 gpurun_out/p4.log; exit 1; }
$P sample_world_1920x1080_s64_l2 --scene sample_world > gpurun_out/p5.log 2>&1 || { tail -5 gpurun_out/p5.log; exit 1; }
$P 3d_linear_1920x1080_s128 --scene standalone3d --bvh 0 --spp 128 > gpurun_out/p6.log 2>&1 || { tail -5 gpurun_out/p6.log; exit 1; }
$P 3d_bvh_1920x1080_s128 --scene standalone3d --bvh 1 --spp 128 > gpurun_out/p7.log 2>&1 || { tail -5 gpurun_out/p7.log; exit 1; }
echo all profiled
