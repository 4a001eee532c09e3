#!/bin/bash
# PMC passes of an arbitrary python script on the GPU box (run through gpurun from the repo root):
#   bash tools/pmc_run_script.sh <tag> <script.py> <args ...>          (environment variables pass through)
# Each pass is its own rocprofv3 run: --pmc <counters> with --kernel-trace and nothing else (no sys/hip/hsa trace domains), the program
# directly after --, as the pool requires.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
p1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
p2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU"
p3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH TCC_HIT_sum TCC_MISS_sum"
p4="FETCH_SIZE"
p5="WRITE_SIZE"
i=0
for p in "$p1" "$p2" "$p3" "$p4" "$p5"; do
  i=$((i+1))
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $out/pass$i -- python "$@" > $out/pass$i.log 2> $out/pass$i.err || { echo "pass $i failed"; tail -5 $out/pass$i.err; exit 1; }
  echo "pass $i done"
done
python tools/pmc_summary.py $out > $out/summary.csv
find $out -name "*.csv" ! -name summary.csv -delete
