#!/bin/bash
# After tools/profile_all_r3.sh: copies the summaries of all seven workloads into profiles/r03.
set -e
for t in complex_world_1920x1080_s64_l2 complex_world_3840x2160_s64_l2 cornell_box_3840x2160_s16_l16 cornell_box_absorption_3840x2160_s16_l16 sample_world_1920x1080_s64_l2; do
  python tools/collect_profiles.py $t profiles/r03 render2d_kernel > /dev/null
done
for t in 3d_linear_1920x1080_s128 3d_bvh_1920x1080_s128; do
  python tools/collect_profiles.py $t profiles/r03 render3d_kernel > /dev/null
done
ls profiles/r03
