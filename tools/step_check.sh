#!/bin/bash
# Quick loop for work on the one-launch VAE step (GPU box): phase stamps, parity tests, bench lines.
timeout -k 10 120 python tools/vae_stamps.py 4096 7 4 2>&1 | grep -v amdgpu.ids | awk '{printf "%s | ", $0} END {print ""}'
timeout -k 10 300 python -u -m pytest tests/test_gpu_vae.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for cfg in "64 7" "4096 7" "4096 30" "65536 7"; do
set -- $cfg
timeout -k 10 120 python bench.py --batch $1 --cars $2 --no-agents --no-cpu-baseline --windows 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('B=$1 N=$2 ms/step %.4f  Mlanes/s %.1f  step frac %.3f  kernels %s' % (d['ms_per_step'], d['value']/1e6, d['step_frac_of_fp32_matrix_peak'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()}))"
done
