#!/bin/bash
# Usage (GPU box): tools/pmc_quick.sh <tag> [bench args]: two PMC passes over k_screen_quad only.
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcq_$tag; raw=/tmp/pmcq_raw_$tag
rm -rf $raw; mkdir -p $out $raw; cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE FETCH_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "k_screen_quad" --output-format csv -d $raw/pmc$i -o pmc -- python $root/bench.py "$@" > $out/bench_pmc$i.log 2>&1
done
python $root/tools/prof_summary.py $raw $out
grep -A30 "k_screen_quad" $out/pmc_summary.txt
