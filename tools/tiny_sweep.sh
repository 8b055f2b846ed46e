#!/bin/bash
# HATD3Triple / WorkerTD3 update time against the tile-count limit of the 16 x 16 product kernels
for lim in 0 32 64 128 256 512; do
  echo "HHRL_GEMM_TINY=$lim"
  HHRL_GEMM_TINY=$lim timeout -k 10 100 python tools/td3_trace.py triple 64 300 2>/dev/null | tail -1
  HHRL_GEMM_TINY=$lim timeout -k 10 100 python tools/td3_trace.py triple 1024 60 2>/dev/null | tail -1
done
