#!/bin/bash
# Runs the round's gpurun calls one after the other as soon as the pool accepts them (tools/gpu_poll.sh retries each while it is
# refused): the verified-tree run, the profile collection, then the A/B session.  Logs: gpurun_out/r06_chain_<stage>.log
cd "$(dirname "$0")/.."
for stage in ${@:-first profiles ab1 ab2}; do
    git rev-parse HEAD > tools/.head 2>/dev/null
    case $stage in
        first) t=2400;; profiles) t=3000;; ab1|ab2) t=3300;; ab) t=7200;; *) t=1800;;
    esac
    tools/gpu_poll.sh $t tools/gpu_session_r06.sh $stage > gpurun_out/r06_chain_$stage.log 2>&1
    echo "stage $stage: rc=$? $(date -u +%T)" >> gpurun_out/r06_chain.log
done
