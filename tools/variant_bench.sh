for lib in libsdsm_hip.so; do
  for wl in bbbc039_like gowt1_like synthetic4096; do
    SDSM_HIP_LIB=superdsm_amd/$lib timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-extras --min-gpu-seconds 0.3 > gpurun_out/v_${lib}_$wl.json 2> gpurun_out/v_${lib}_$wl.err
    python -c "
import json,sys; d=json.load(open('gpurun_out/v_${lib}_$wl.json')); print('$lib $wl', round(d['value']), 'ms/step %.2f'%d['ms_per_step'], 'solve %.2f'%d['roofline']['kernel_ms'], 'setup %.2f'%d['roofline']['setup_kernel_ms'], d['status_counts'])"
  done
done
