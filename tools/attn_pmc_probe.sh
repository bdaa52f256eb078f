#!/bin/bash
# counter passes over tools/attn_pmc_probe.py (one --pmc set per run, never combined with traces)
out=gpurun_out/attn_pmc; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/mm-vqa-healthcare_amd
i=0
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/$out/p$i -o p -- python $R/tools/attn_pmc_probe.py > $R/$out/p$i.log 2>&1
  rc=$?
  echo "pass $i ($set) rc=$rc"
  if [ $rc -ge 124 ]; then echo "stopping"; exit 1; fi
  f=$(find $R/$out/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python $R/tools/attn_pmc_probe.py $f > $R/$out/p$i.txt 2>&1; rm -f $f; fi
done
cat $R/$out/p*.txt
