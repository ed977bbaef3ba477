#!/bin/bash
# Runner for a list of GPU steps inside ONE gpurun call: every step under its own `timeout -k 10`, log per step under gpurun_out/<tag>/;
# a step that fails with an ordinary error lets the next one run, a step that TIMES OUT or is killed ends the call (a hung GPU must
# not be given more work).   usage: source tools/gpu_steps.sh <tag>;  step <name> <seconds> <command...>
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift; shift
  echo "== $name: $*" | tee -a $OUT/steps.log
  local t0=$(date +%s)
  timeout -k 10 $secs "$@" > $OUT/$name.log 2> $OUT/$name.err
  local rc=$?
  echo "== $name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $OUT/steps.log
  tail -3 $OUT/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping" | tee -a $OUT/steps.log; tail -5 $OUT/$name.err; exit 1; fi
  if [ $rc -ne 0 ]; then tail -15 $OUT/$name.err; fi
  return 0
}
