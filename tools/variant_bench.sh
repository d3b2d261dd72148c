# Diagnostic: bench lines of alternative builds (superdsm_amd/libsdsm_hip_<name>.so) on the workloads of the bench
# usage: bash tools/variant_bench.sh lib1.so lib2.so ...   (WLS="bbbc039_like|bbbc039_like --same-layout|synthetic4096" to choose)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
IFS='|' read -ra wls <<< "${WLS:-bbbc039_like|bbbc039_like --same-layout|gowt1_like|nih3t3_like|synthetic4096}"
for lib in "$@"; do
  for wl in "${wls[@]}"; do
    tag=$(echo "$wl" | tr -d ' -')
    SDSM_HIP_LIB=superdsm_amd/$lib timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-extras --min-gpu-seconds 0.5 > gpurun_out/v_${lib}_$tag.json 2> gpurun_out/v_${lib}_$tag.err
    python -c "
import json,sys; d=json.load(open('gpurun_out/v_${lib}_$tag.json')); print('$lib $wl', round(d['value']), 'ms/step %.2f'%d['ms_per_step'], 'solve %.2f'%d['roofline']['kernel_ms'], 'setup %.2f'%d['roofline']['setup_kernel_ms'], d['status_counts'])"
  done
done
