#!/bin/bash
# After `bash tools/run_profiles.sh <tag> a` and `... b` on the GPU box: the committed summaries under profiles/ from what gpurun merged into gpurun_out/.   usage: bash tools/collect_profiles.sh r04
set -e
tag=${1:-r04}
cd "$(dirname "$0")/.."
python3 tools/summarize_profiles.py $tag bbbc039_like 8 > /dev/null
python3 tools/summarize_profiles.py ${tag}_same bbbc039_like_same 8 > /dev/null
python3 tools/summarize_profiles.py ${tag}_gowt1 gowt1_like 1 > /dev/null
python3 tools/summarize_profiles.py ${tag}_s4096 synthetic4096 1 > /dev/null
python3 tools/kernel_timeline.py $tag > /dev/null
cp gpurun_out/${tag}_class_stats_bbbc039_8.log profiles/${tag}_class_stats_bbbc039_8images.txt
cp gpurun_out/${tag}_class_stats_bbbc039_8same.log profiles/${tag}_class_stats_bbbc039_8copies.txt
cp gpurun_out/${tag}_class_stats_gowt1.log profiles/${tag}_class_stats_gowt1.txt
cp gpurun_out/${tag}_class_stats_nih3t3.log profiles/${tag}_class_stats_nih3t3.txt
cp gpurun_out/${tag}_class_stats_s4096.log profiles/${tag}_class_stats_synthetic4096.txt
cp gpurun_out/${tag}_phase.log profiles/${tag}_phase_cycles.txt
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
cp gpurun_out/${tag}_bench_image_set.json profiles/${tag}_bench_image_set.json
cp gpurun_out/${tag}_bench_2rank.json profiles/${tag}_bench_2rank_one_card.json
cp gpurun_out/${tag}_bench_sharded_2rank.json profiles/${tag}_bench_sharded_2rank_one_card.json
python3 - <<P
import json
d = json.loads(open('profiles/${tag}_bench.json').read().strip().splitlines()[-1])
print('bench line: %.0f %s, %.2f ms per step, roofline frac %.4f, traffic %s' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']))
P
