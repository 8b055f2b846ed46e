#!/bin/bash
# one-launch (HHRL_VAE_STEP=1) against four-launch (=4) VAE step for small batches and growing lane lengths
for cfg in "64 1" "64 4" "64 7" "64 12" "64 16" "64 24" "64 32" "64 40" "256 16" "256 40" "1024 16" "1024 40" "2048 40"; do
set -- $cfg
for form in 1 4; do
HHRL_VAE_STEP=$form timeout -k 10 120 python bench.py --batch $1 --cars $2 --no-agents --no-cpu-baseline --windows 2 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('form $form B=$1 N=$2 ms/step %.4f' % d['ms_per_step'])"
done; done
