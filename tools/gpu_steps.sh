#!/bin/bash
# Runs GPU steps one after another, each under its own time limit, and stops at the first step that ends in
# anything but success or an ordinary failure (status 1: a failed assertion): after a fault, an abort, a
# timeout or a kill nothing more is started on the GPU.
#   step <seconds> <log> <command...>
step() {
  local limit=$1 log=$2; shift 2
  mkdir -p "$(dirname "$log")"
  echo "== $(date +%H:%M:%S) $* (limit ${limit}s) -> $log"
  timeout -k 10 "$limit" "$@" > "$log" 2>&1
  local rc=$?
  echo "   rc=$rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then
    echo "STOP: step ended with status $rc; tail of $log:"; tail -30 "$log"; exit $rc
  fi
  return 0
}
