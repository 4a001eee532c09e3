#!/bin/bash
# Kernel trace of ONE bench configuration on the GPU box (run through gpurun from the repo root):
#   bash tools/trace_one.sh <tag> <name> <bench args ...>
# rocprofv3 --kernel-trace --stats only, the program directly after --; leaves gpurun_out/<tag>/<name>_kernel_stats.csv
set -e
tag=$1; name=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python bench.py "$@" --no-cpu-baseline --no-single-solve --no-extras --no-legs > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }
f=$(find $out/$name -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${name}_kernel_stats.csv
rm -rf $out/$name
cut -c1-110 $out/${name}_kernel_stats.csv | head -16
