#!/bin/bash
# Diagnostic: bench lines for a list of (environment, library) settings.   usage: bash tools/sweep_bench.sh "label;VAR=1 VAR2=2;lib.so" ...
# (WLS="bbbc039_like|bbbc039_like --same-layout|synthetic4096" chooses the workloads; lib defaults to libsdsm_hip.so)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
IFS='|' read -ra wls <<< "${WLS:-bbbc039_like|bbbc039_like --same-layout|gowt1_like|nih3t3_like|synthetic4096}"
for spec in "$@"; do
  IFS=';' read -r label envs lib <<< "$spec"
  lib=${lib:-libsdsm_hip.so}
  for wl in "${wls[@]}"; do
    tag=$(echo "${label}_$wl" | tr -d ' -')
    env $envs SDSM_HIP_LIB=superdsm_amd/$lib timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-extras --min-gpu-seconds ${MINGPU:-0.5} > gpurun_out/s_$tag.json 2> gpurun_out/s_$tag.err
    python -c "
import json; d=json.load(open('gpurun_out/s_$tag.json')); print('$label | $wl |', round(d['value']), 'ms/step %.2f'%d['ms_per_step'], 'solve %.2f'%d['roofline']['kernel_ms'], 'setup %.2f'%d['roofline']['setup_kernel_ms'], d['status_counts'])" 2>&1 | tail -1
  done
done
