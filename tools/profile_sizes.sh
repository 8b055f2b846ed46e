#!/bin/bash
# Kernel-time pass of the VAE step at sizes other than the headline one (rocprofv3 --kernel-trace --stats).
# usage: tools/profile_sizes.sh <tag> "<lanes> <cars> <latent>" ...     (run from the repo root on a GPU box)
tag=$1; shift
export TMPDIR=/tmp
for shape in "$@"; do
  set -- $shape
  out=$(pwd)/gpurun_out/prof_${tag}_$1x$2
  mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 100 --warmup 10 \
    --batch $1 --cars $2 --latent $3 --no-cpu-baseline --no-agents > $out/bench_under_trace.json || exit 1
  cp $(find $out/trace -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_$1x$2_kernel_stats.csv
done
echo done
