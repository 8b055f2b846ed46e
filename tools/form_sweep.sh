for form in 1 4; do
for cfg in "64 7" "1024 7" "4096 7" "4096 30" "16384 7" "65536 7"; do
set -- $cfg
HHRL_VAE_STEP=$form timeout -k 10 120 python bench.py --batch $1 --cars $2 --no-agents --no-cpu-baseline --windows 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('form $form B=$1 N=$2 ms/step %.4f  Mlanes/s %.1f  step frac %.3f  kernels %s' % (d['ms_per_step'], d['value']/1e6, d['step_frac_of_fp32_matrix_peak'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()}))"
done; done
