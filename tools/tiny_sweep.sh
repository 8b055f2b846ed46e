#!/bin/bash
# HATD3Triple update time against the tile-count limit of the 16 x 16 product kernels
for lim in ${LIMITS:-0 32 64 128 256 512}; do
  echo "HHRL_GEMM_TINY=$lim"
  HHRL_GEMM_TINY=$lim timeout -k 10 100 python tools/td3_trace.py triple 64 300 2>/dev/null | tail -1
done
