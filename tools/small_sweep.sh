#!/bin/bash
# agent update time against the tile-count limit of the small-product kernels (beyond it: the 64x64 / 128x128 tile kernel)
for lim in ${LIMITS:-512 256 128 64}; do
  echo "HHRL_GEMM_SMALL_MAX=$lim"
  HHRL_GEMM_SMALL_MAX=$lim timeout -k 10 100 python tools/td3_trace.py triple ${BATCH:-1024} 60 2>/dev/null | tail -1
done
