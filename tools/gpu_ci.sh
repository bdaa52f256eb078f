#!/bin/bash
# Run GPU test stages in order on the GPU box; stop at the first stage that times out or is killed (never retry).
# usage: tools/gpu_ci.sh stage1 stage2 ...   (stages: ops model smoke bench prof)
mkdir -p gpurun_out
run() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/ci.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/ci.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then
    echo "!!! $name was killed/aborted (rc=$rc): stopping" | tee -a gpurun_out/ci.log
    exit $rc
  fi
  return 0
}
: > gpurun_out/ci.log
for st in "$@"; do
  case $st in
    selftest) run selftest 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k selftest ;;
    ops)      run ops 900 python -m pytest tests/test_gpu_ops.py -m gpu -q ;;
    model)    run model 900 python -m pytest tests/test_gpu_model.py -m gpu -q ;;
    smoke)    run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)    run bench 900 python bench.py ;;
    all)      run all 1100 python -m pytest tests -m gpu -q -x ;;
    *) echo "unknown stage $st" ;;
  esac
done
