#!/bin/bash
# update / step times with kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) against the runtime's default
for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 100 python tools/td3_trace.py triple 64 300 2>/dev/null | tail -1
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 100 python tools/td3_trace.py worker 64 500 2>/dev/null | tail -1
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python bench.py --no-agents --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vae ms/step', d['ms_per_step'], d['windows'])"
done
