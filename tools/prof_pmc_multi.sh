#!/bin/bash
# usage: tools/prof_pmc_multi.sh <tag> "<COUNTER1 COUNTER2 ...>" <command...>
# One rocprofv3 --pmc pass with SEVERAL counters of one hardware block budget (<= 8 SQ, <= 4 TCC ...; --kernel-trace only, never
# combined with other trace domains); per-kernel means per launch -> gpurun_out/<tag>.csv (tools/pmc_summary.py)
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
( cd $root && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o $tag -- "$@" ) > $root/gpurun_out/$tag.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python $root/tools/pmc_summary.py "$f" $root/gpurun_out/$tag.csv
rm -rf $out
