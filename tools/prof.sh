#!/bin/bash
# Usage (on the GPU box): tools/prof.sh <tag> [bench args...]
# 1) kernel trace + stats, 2) PMC passes (separate runs; counters never mixed with trace domains).
# PROF_TRACE_ONLY=1 stops after the kernel trace.  Always --output-format csv: the default rocpd database makes
# --stats post-processing take tens of minutes on the data generator's many small launches.
# Raw rocprofv3 output stays in /tmp; only the small summaries land in gpurun_out/prof_<tag>/.
tag=$1; shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
raw=/tmp/prof_raw_$tag
rm -rf $raw; mkdir -p $out $raw
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/trace -o trace -- python $root/bench.py "$@" > $out/bench_trace.log 2>&1
find $raw/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
if [ -n "$PROF_TRACE_ONLY" ]; then head -14 $out/kernel_stats.csv | cut -c1-160; exit 0; fi
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "k_assign|k_accumulate|k_combine|k_scatter|k_fwht|k_plan|k_reduce|k_finalize|k_prep|k_screen|k_exact|k_hist|k_point|k_bounds|k_copy|k_center|k_build" --output-format csv -d $raw/pmc$i -o pmc -- python $root/bench.py "$@" > $out/bench_pmc$i.log 2>&1
done
python $root/tools/prof_summary.py $raw $out
ls -la $out
