#!/bin/bash
# Kernel-time, HBM-traffic and matrix-core passes of the headline bench under rocprofv3.
# Each pass is a run of its own (counters never share a run with a trace).
# usage: tools/profile_bench.sh <tag>     (run from the repo root on a GPU box)
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-agents"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $args > $out/bench_under_trace.json
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $args > /dev/null
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $args > /dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 $args > /dev/null
echo done
