#!/bin/bash
# A/B of two builds of the library on the four agent-update configurations of bench.py (tools/td3_trace.py, wall time per update).
# usage: tools/td3_ab.sh <libA.so> <libB.so> [rounds]      (run from the repo root on a GPU box; "-" = the default library)
a=$1; b=$2; rounds=${3:-2}
out=gpurun_out/td3_ab.txt
: > $out
for r in $(seq $rounds); do
  for v in "$a" "$b"; do
    if [ "$v" = "-" ]; then unset HHRL_HIP_LIB; else export HHRL_HIP_LIB=$PWD/$v; fi
    for c in "worker 64" "worker 4096" "triple 64" "triple 1024"; do
      set -- $c
      echo -n "$v " >> $out
      timeout -k 10 200 python tools/td3_trace.py $1 $2 300 2>/dev/null | cut -c1-110 >> $out || exit 1
    done
  done
done
cat $out
