#!/bin/bash
# Kernel traces of the bench configurations on the GPU box (run through gpurun from the repo root):
#   bash tools/trace_run.sh <tag>
# rocprofv3 --kernel-trace --stats only, the program directly after --, no CPU pool under the profiler (--no-cpu-baseline).
# Leaves gpurun_out/<tag>/{C2,C2_one_group,C4,C5}_kernel_stats.csv and the per-(kernel, grid) medians of the C2 trace.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
common="--no-cpu-baseline --no-single-solve --no-extras --no-legs"
run() {  # name, env..., -- bench args
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }
  f=$(find $out/$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/${name}_kernel_stats.csv
  echo "$name done: $(head -c 300 $out/$name.json | cut -c1-160)"
}
run C2 --steps 3 --warmup 2 $common
python tools/trace_medians.py $out/C2 > $out/C2_medians.txt 2>&1 || true
EGDST_GROUPS=1 EGDST_ADAPTIVE=0 run C2_one_group --steps 2 --warmup 1 $common
run C4 --scaling strong --workload C4 --ndraw-total 32 --steps 3 --warmup 2 $common
run C5 --scaling strong --workload C5 --ndraw-total 128 --steps 2 --warmup 1 $common
# keep the summaries only (the traces are tens of MB)
for n in C2 C2_one_group C4 C5; do rm -rf $out/$n; done
ls -la $out
