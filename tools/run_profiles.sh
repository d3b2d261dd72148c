#!/bin/bash
# Every profile of a round in one call on the GPU box:   bash tools/run_profiles.sh r03
# (rocprofv3 passes of the default bench workload -- kernel trace + statistics, FETCH_SIZE, WRITE_SIZE, SQ counters, each in its
# own run --, kernel statistics + FETCH / WRITE of the GOWT1-like and the synthetic 4096^2 launch, the phase counters of the diagnostic
# build, the bench lines).  Outputs under gpurun_out/<tag>_*; summaries are made afterwards, off the box, by
# tools/summarize_profiles.py <tag> and tools/kernel_timeline.py <tag>.
set -e
tag=${1:-r04}
part=${2:-ab}     # a: the rocprofv3 passes; b: diagnostic build, multi-rank rehearsals, the bench line (two calls fit gpurun's time limit)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu --no-extras --steps 6 --warmup 2 --repeats 2 --min-gpu-seconds 0"
if [[ $part == *a* ]]; then
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_gowt1_* gpurun_out/${tag}_s4096_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- $B > gpurun_out/${tag}_p1.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- $B > gpurun_out/${tag}_p2.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- $B > gpurun_out/${tag}_p3.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_sq -- $B > gpurun_out/${tag}_p4.log 2>&1
echo "sq done"
# the round-2 step (8 copies of one image) for comparison: kernel statistics + FETCH
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_same_trace -- $B --same-layout > gpurun_out/${tag}_p5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_same_fetch -- $B --same-layout > gpurun_out/${tag}_p6.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_same_sq -- $B --same-layout > gpurun_out/${tag}_p7.log 2>&1
echo "same-layout done"
for wl in gowt1_like synthetic4096; do
    short=${wl/_like/}; short=${short/synthetic/s}
    W="python3 bench.py --workload $wl --no-cpu --no-extras --steps 3 --warmup 1 --repeats 1 --min-gpu-seconds 0"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${short}_trace -- $W > gpurun_out/${tag}_${short}_p1.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_${short}_fetch -- $W > gpurun_out/${tag}_${short}_p2.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_${short}_write -- $W > gpurun_out/${tag}_${short}_p3.log 2>&1
    echo "$wl done"
done
fi
if [[ $part == *b* ]]; then
if [ -f superdsm_amd/libsdsm_hip_prof.so ]; then
    SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so timeout -k 10 300 python3 tools/gpu_phase_profile.py > gpurun_out/${tag}_phase.log 2>&1
    SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so timeout -k 10 300 python3 tools/class_stats.py bbbc039_like 8 > gpurun_out/${tag}_class_stats_bbbc039_8.log 2>&1
    SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so timeout -k 10 300 python3 tools/class_stats.py synthetic4096 > gpurun_out/${tag}_class_stats_s4096.log 2>&1
    SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so timeout -k 10 300 python3 tools/class_stats.py gowt1_like > gpurun_out/${tag}_class_stats_gowt1.log 2>&1 || true
    SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so timeout -k 10 300 python3 tools/class_stats.py nih3t3_like > gpurun_out/${tag}_class_stats_nih3t3.log 2>&1 || true
    SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so timeout -k 10 300 python3 tools/class_stats.py bbbc039_like 8 same > gpurun_out/${tag}_class_stats_bbbc039_8same.log 2>&1 || true
    echo "phase done"
fi
python3 bench.py --mode image_set --images 3 > gpurun_out/${tag}_bench_image_set.json 2> gpurun_out/${tag}_bench_image_set.err
echo "image set done"
python3 bench.py --gpus 2 --steps 4 --warmup 2 --no-extras --no-cpu > gpurun_out/${tag}_bench_2rank.json 2> gpurun_out/${tag}_bench_2rank.err
echo "2 ranks (one card) done"
python3 bench.py --gpus 2 --mode sharded --steps 3 --warmup 1 > gpurun_out/${tag}_bench_sharded_2rank.json 2> gpurun_out/${tag}_bench_sharded_2rank.err
echo "sharded, 2 ranks (one card) done"
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done"
fi
