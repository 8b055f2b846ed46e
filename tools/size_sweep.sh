#!/bin/bash
# Size sweep of the fused VAE step: one bench line per shape into gpurun_out/sweep_<tag>.jsonl
# usage: tools/size_sweep.sh <tag>     (run from the repo root on a GPU box)
tag=${1:-sweep}
out=gpurun_out/sweep_$tag.jsonl
mkdir -p gpurun_out
: > $out
for shape in "64 7 4" "1024 7 4" "4096 7 2" "4096 7 4" "4096 7 16" "4096 30 4" "16384 7 4" "65536 7 4"; do
  set -- $shape
  timeout -k 10 200 python bench.py --batch $1 --cars $2 --latent $3 --no-agents --no-cpu-baseline 2>/dev/null | grep '^{' >> $out || exit 1
done
echo done
