#!/bin/bash
# usage: tools/prof_trace.sh <tag> [bench.py args...]
# rocprofv3 --kernel-trace --stats of `python bench.py <args>` on the GPU box; writes gpurun_out/<tag>_kernel_stats.csv
# (per-kernel calls / total / average, via tools/rocpd_stats.py) next to the bench line gpurun_out/<tag>.json
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o $tag -- python $root/bench.py "$@" > $root/gpurun_out/$tag.json 2> $root/gpurun_out/$tag.err
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python $root/tools/rocpd_stats.py "$db" $root/gpurun_out/${tag}_kernel_stats.csv
rm -rf $root/gpurun_out/prof_$tag
