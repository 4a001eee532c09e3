#!/bin/bash
# Kernel trace of a few C2 batch solves and its gap / concurrency analysis (run through gpurun from the repo root):
#   bash tools/trace_gaps_run.sh <tag> [env assignments ...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python tests/diag/gpu_batch_trace.py C2 4096 3 > $out/run.log 2> $out/run.err || { tail -5 $out/run.err; exit 1; }
python tools/trace_gaps.py $out/t 3 > $out/gaps.txt
python tools/trace_queues.py $out/t 3 > $out/queues.txt
rm -rf $out/t
cat $out/gaps.txt
