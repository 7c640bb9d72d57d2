#!/bin/bash
# usage: tools/gpu_poll.sh <gpurun timeout s> <command...>  -- retries a gpurun call every 4 minutes while the pool refuses it (exit 2 / 3)
t=$1; shift
for i in $(seq 1 150); do
    out=$(cd /tmp && /usr/local/graft/bin/gpurun --timeout $t -- "$@" 2>&1); rc=$?
    if ! echo "$out" | grep -q "status=refused\|status=busy\|no box"; then echo "$out" | tail -60; exit $rc; fi
    sleep 240
done
echo "gpu_poll: still refused after 150 tries"; exit 2
