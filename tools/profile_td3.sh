#!/bin/bash
# rocprofv3 kernel-trace pass over the TD3 updates of bench.py's four agent configurations.
# usage: tools/profile_td3.sh <tag>     (run from the repo root on a GPU box; summaries: tools/summarize_prof.py --td3 <tag>)
set -e
tag=${1:-r02}
root=$(pwd)
export TMPDIR=/tmp
for cfg in "worker 64" "worker 4096" "triple 64" "triple 1024"; do
  set -- $cfg
  out=$root/gpurun_out/prof_${tag}_td3_$1_b$2
  mkdir -p $out
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/td3_trace.py $1 $2 200 > $out/line.json
  cat $out/line.json
done
echo done
