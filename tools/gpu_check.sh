#!/bin/bash
# One GPU session: the GPU tests, the work-in-progress no-alloc test, a bench run.  Stops at the first
# step that crashes, times out or hangs (exit status other than 0 / 1); assertion failures go on.
#   tools/gpu_check.sh <tag> [bench args...]
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
step() { # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name"
  timeout -k 10 $secs "$@" > $OUT/${TAG}_$name.log 2>&1
  local rc=$?
  echo "   rc=$rc"; tail -n 6 $OUT/${TAG}_$name.log
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name (rc=$rc)"; exit $rc; fi
}
step tests 1000 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread -s
if [ -f tests/wip_no_alloc.py ]; then step noalloc 300 python -u -m pytest tests/wip_no_alloc.py -v --timeout 120 --timeout-method thread -s; fi
step bench 400 python -u bench.py "$@"
