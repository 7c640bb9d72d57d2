#!/bin/bash
# usage (through gpurun): tools/gpu_session_r06.sh [first|ab|ab1|ab2|profiles]
#   first    = the verified-tree run of round 6: whole GPU suite + smoke() + the default bench line, logs kept
#   ab       = tools/next_gpu_session.sh (the prepared A/Bs; needs the tools/_var libraries built before the call)
#   profiles = tools/collect_profiles.sh r06
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=gpurun_out; mkdir -p $o
what=${1:-first}
case $what in
first)
    cat tools/.head > $o/r06_first_head.txt 2>/dev/null
    (timeout 1200 python -m pytest tests -m gpu -x -q > $o/r06_gputest_first.log 2>&1; echo "pytest rc=$?" >> $o/r06_gputest_first.log)
    grep -E "passed|failed|error|rc=" $o/r06_gputest_first.log | tail -4
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $o/r06_smoke_first.log 2>&1; echo "smoke rc=$?" >> $o/r06_smoke_first.log)
    tail -2 $o/r06_smoke_first.log
    timeout 900 python bench.py > $o/r06_bench_first.json 2> $o/r06_bench_first.err; tail -c 600 $o/r06_bench_first.json
    ;;
ab) shift; tools/next_gpu_session.sh "$@" ;;
ab1) STAGES="6 0 1 2" tools/next_gpu_session.sh ;;      # the A/B session in two calls of about 45 minutes each
ab2) STAGES="3 4 5" tools/next_gpu_session.sh ;;
profiles) tools/collect_profiles.sh r06 ;;
esac
