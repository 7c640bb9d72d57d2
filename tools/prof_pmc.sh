#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <COUNTER> <command...>
# One rocprofv3 --pmc pass (its own run: counters are never combined with the stats / trace passes) of <command>;
# per-kernel means per launch -> gpurun_out/<tag>_<counter>.csv (tools/pmc_summary.py)
tag=$1; ctr=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_${tag}_$ctr
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
( cd $root && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $tag -- "$@" ) > $root/gpurun_out/${tag}_${ctr}.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python $root/tools/pmc_summary.py "$f" $root/gpurun_out/${tag}_$(echo $ctr | tr 'A-Z' 'a-z').csv
rm -rf $out
