#!/bin/bash
# Secondary bench modes (DESIGN.md "Other modes"): one JSON line each under gpurun_out/modes/.
mkdir -p gpurun_out/modes
run() {  # name args...
  local name=$1; shift
  echo "=== $name"
  timeout -k 10 420 python bench.py --no-cpu-baseline --no-roofline --no-secondary "$@" > gpurun_out/modes/$name.log 2>&1
  local rc=$?
  grep '^{' gpurun_out/modes/$name.log | tail -1 > gpurun_out/modes/$name.json
  echo "=== $name rc=$rc $(python -c "import json,sys; d=json.load(open('gpurun_out/modes/$name.json')); print(d['value'], d['unit'], d['ms_per_step'], 'ms')" 2>/dev/null)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
run cls_B64 --batch 64
run t5small_B64 --head t5 --batch 64
run t5base_B64 --head t5 --t5 t5-base --batch 64
run pretrain_B128 --head pretrain --batch 128 --steps 4 --warmup 2
run large_cls_B32 --arch large --batch 32 --steps 3 --warmup 2
run large_t5large_B16 --arch large --head t5 --t5 t5-large --batch 16 --steps 3 --warmup 2
