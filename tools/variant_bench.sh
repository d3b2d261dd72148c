# Diagnostic: bench lines of alternative builds (superdsm_amd/libsdsm_hip_<name>.so), 8 different layouts and the round-2 step (8 copies)
for lib in "$@"; do
  for mode in "" "--same-layout"; do
    SDSM_HIP_LIB=superdsm_amd/$lib timeout -k 10 200 python bench.py $mode --no-cpu --no-extras --min-gpu-seconds 0.5 > gpurun_out/v_${lib}_$mode.json 2> gpurun_out/v_${lib}_$mode.err
    python -c "
import json,sys; d=json.load(open('gpurun_out/v_${lib}_$mode.json')); print('$lib $mode', round(d['value']), 'ms/step %.2f'%d['ms_per_step'], 'solve %.2f'%d['roofline']['kernel_ms'], 'setup %.2f'%d['roofline']['setup_kernel_ms'], d['status_counts'])"
  done
done
