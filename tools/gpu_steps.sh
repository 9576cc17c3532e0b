#!/bin/bash
# Runs GPU steps one after the other on a gpurun box: step <seconds> <log> <command...>.  A step that is killed at its limit ends the
# whole call (no further GPU step behind a hung one); a step that merely fails is logged and the next one runs.
O=${O:-gpurun_out/r6}; mkdir -p $O
step() {
  local lim=$1 log=$2; shift 2
  echo "== $(date +%T) $* (limit ${lim}s) -> $log"
  timeout -k 10 $lim "$@" > $O/$log 2> $O/${log%.*}.err
  local rc=$?
  echo "   rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its limit: stopping"; exit 124; fi
  return 0
}
