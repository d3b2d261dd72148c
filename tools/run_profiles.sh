#!/bin/bash
# Every profile of a round in one call on the GPU box:   bash tools/run_profiles.sh r02
# (rocprofv3 passes of the default bench workload -- kernel trace + statistics, FETCH_SIZE, WRITE_SIZE, SQ counters, each in its
# own run --, the phase counters of the diagnostic build, the bench lines).  Outputs under gpurun_out/<tag>_*; summaries are made
# afterwards, off the box, by tools/summarize_profiles.py <tag> and tools/kernel_timeline.py <tag>.
set -e
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --no-cpu --no-extras --steps 6 --warmup 2 --repeats 2 > gpurun_out/${tag}_p1.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --no-cpu --no-extras --steps 6 --warmup 2 --repeats 2 > gpurun_out/${tag}_p2.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --no-cpu --no-extras --steps 6 --warmup 2 --repeats 2 > gpurun_out/${tag}_p3.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_sq -- python3 bench.py --no-cpu --no-extras --steps 6 --warmup 2 --repeats 2 > gpurun_out/${tag}_p4.log 2>&1
echo "sq done"
if [ -f superdsm_amd/libsdsm_hip_prof.so ]; then
    SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so timeout -k 10 300 python3 tools/gpu_phase_profile.py > gpurun_out/${tag}_phase.log 2>&1
    echo "phase done"
fi
python3 bench.py --mode image_set --images 3 > gpurun_out/${tag}_bench_image_set.json 2> gpurun_out/${tag}_bench_image_set.err
echo "image set done"
python3 bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/${tag}_bench_2rank.json 2> gpurun_out/${tag}_bench_2rank.err
echo "2 ranks (one card) done"
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done"
