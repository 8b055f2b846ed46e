#!/bin/bash
# rocprofv3 passes over the TD3 updates of bench.py's four agent configurations: kernel trace + stats, then the
# counter passes, each a run of its own (counters never share a run with a trace): HBM read and write bytes,
# matrix-pipe busy cycles and executed matrix operations.
# usage: tools/profile_td3.sh <tag> [updates]   (run from the repo root on a GPU box; summaries: tools/summarize_prof.py --td3 <tag>)
set -e
tag=${1:-r03}
updates=${2:-200}
root=$(pwd)
export TMPDIR=/tmp
for cfg in "worker 64" "worker 4096" "triple 64" "triple 1024"; do
  set -- $cfg
  out=$root/gpurun_out/prof_${tag}_td3_$1_b$2
  mkdir -p $out
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/td3_trace.py $1 $2 $updates > $out/line.json
  cat $out/line.json
  find $out/trace -name "*_kernel_trace.csv" -delete        # the per-dispatch rows (tens of MB); the stats file is what is kept
  export TD3_TRACE_BRIEF=1                                   # counter passes: no half second of clock warm-up, nothing is timed
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/td3_trace.py $1 $2 40 > /dev/null
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/td3_trace.py $1 $2 40 > /dev/null
  timeout -s KILL 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 tools/td3_trace.py $1 $2 40 > /dev/null
  unset TD3_TRACE_BRIEF
  echo "$1 $2 done"
done
echo done
