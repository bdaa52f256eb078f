#!/bin/bash
# Round profile on the GPU box: the default bench line, rocprofv3 kernel statistics of the same workload, and three
# PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA-busy + GRBM clock; separate runs, --pmc never combined with other traces).
# Stops at the first step that is killed.  Outputs under gpurun_out/prof_$1/.
tag=${1:-final}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $out/steps.log
  timeout -k 10 "$to" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a $out/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then exit $rc; fi
}
: > $out/steps.log
step bench 600 python bench.py
grep '^{' $out/bench.log | tail -1 > $out/bench_line.json
step stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python bench.py --no-cpu-baseline --no-roofline --no-secondary --no-parity --steps 5 --warmup 3
step fetch 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc -o fetch -- python bench.py --no-cpu-baseline --no-roofline --no-secondary --no-parity --steps 2 --warmup 1
step write 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc -o write -- python bench.py --no-cpu-baseline --no-roofline --no-secondary --no-parity --steps 2 --warmup 1
step mfma 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -o mfma -- python bench.py --no-cpu-baseline --no-roofline --no-secondary --no-parity --steps 2 --warmup 1
# the fused cross-attention forward alone (north_star's kernel): kernel times and fabric-side bytes per layer pair
step xattn_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/xattn_stats -o run -- python tools/xattn_pair.py
step xattn_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc -o xfetch -- python tools/xattn_pair.py
step xattn_write 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc -o xwrite -- python tools/xattn_pair.py
python tools/pmc_traffic.py $(find $out/pmc -name 'xfetch_counter_collection.csv' | head -1) $(find $out/pmc -name 'xwrite_counter_collection.csv' | head -1) --pairs 4 > $out/xattn_pmc_traffic.txt 2>&1
# keep what is judged small: per-kernel stats, the counter tables reduced by the tools
find $out -name '*_kernel_stats.csv' -o -name '*_domain_stats.csv' | head
python tools/pmc_traffic.py $(find $out/pmc -name 'fetch_counter_collection.csv' | head -1) $(find $out/pmc -name 'write_counter_collection.csv' | head -1) --json $out/pmc_traffic.json 256 cls > $out/pmc_traffic.txt 2>&1
python tools/pmc_mfma.py $(find $out/pmc -name 'mfma_counter_collection.csv' | head -1) $out/pmc_mfma.json > $out/pmc_mfma.txt 2>&1
rm -f $(find $out -name '*_kernel_trace.csv') $(find $out/pmc -name '*_counter_collection.csv')   # tens of MiB each
ls -la $out
