#!/bin/bash
# usage (on the GPU box, through gpurun): tools/collect_profiles.sh <round tag, e.g. r03>
# Everything profiles/ holds for a round: bench lines of every configuration, kernel-trace stats, the PMC passes
# (FETCH_SIZE / WRITE_SIZE / MfmaUtil, each its own rocprofv3 run) and the FETCH_SIZE calibration.  Output: gpurun_out/<tag>_*
r=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=gpurun_out
mkdir -p $o
# HEAD the snapshot was taken from (written by the caller before gpurun: `git rev-parse HEAD > tools/.head`; .git does not travel)
head=$(cat tools/.head 2>/dev/null || echo unknown)
src=$(cat bench.py video-description-with-spatial-temporal-attention_amd/csrc/*.hip video-description-with-spatial-temporal-attention_amd/csrc/*.cpp video-description-with-spatial-temporal-attention_amd/csrc/*.h | sha256sum | cut -c1-16)
# per-kernel ISA hashes of the sources this library was built from (written BEFORE the call, here hipcc time is GPU time: `tools/isa_stamp.py --out profiles/${r}_ISA.json`);
# `tools/isa_stamp.py --check profiles/${r}_ISA.json` on any later tree says mechanically whether these profiles still describe its kernels
isa=$(sha256sum profiles/${r}_ISA.json 2>/dev/null | cut -c1-16)
echo "{\"round\": \"$r\", \"git_head\": \"$head\", \"sources_sha16\": \"$src\", \"isa_stamp\": \"profiles/${r}_ISA.json\", \"isa_stamp_sha16\": \"$isa\", \"collected\": \"$(date -u +%FT%TZ)\"}" > $o/${r}_STAMP.json
python bench.py > $o/${r}_bench_c2_train.json 2> $o/${r}_bench_c2_train.err
python bench.py --mode forward > $o/${r}_bench_c2_forward.json 2>> $o/${r}_bench.err
python bench.py --config c2v20k --no-cpu-baseline > $o/${r}_bench_c2v20k_train.json 2>> $o/${r}_bench.err
python bench.py --config c4 --precision bf16 --no-cpu-baseline > $o/${r}_bench_c4_train_bf16.json 2>> $o/${r}_bench.err
python bench.py --config c4 --no-cpu-baseline --no-split > $o/${r}_bench_c4_train_fp32.json 2>> $o/${r}_bench.err
python bench.py --mode beam --config c5 --steps 5 --warmup 1 > $o/${r}_bench_c5_beam.json 2>> $o/${r}_bench.err
python bench.py --mode beam --config c1 --beam 1 --steps 50 --warmup 3 > $o/${r}_bench_c1_greedy_beam1.json 2>> $o/${r}_bench.err
python bench.py --mode beam --config c1 --beam 5 --steps 50 --warmup 3 > $o/${r}_bench_c1_beam5.json 2>> $o/${r}_bench.err
python bench.py --mode decode --config c1 --steps 5 --warmup 1 > $o/${r}_bench_c1_decode.json 2>> $o/${r}_bench.err
python bench.py --mode decode --config c1 --beam 5 --steps 5 --warmup 1 --no-cpu-baseline > $o/${r}_bench_c1_decode_beam5.json 2>> $o/${r}_bench.err
# the reference's evaluation workload on its own (the same leg rides in the default line), and the bf16 LDS tile sweep of configs[3]
python bench.py --mode eval > $o/${r}_bench_eval_msvd.json 2>> $o/${r}_bench.err
python tools/gemm_bf16_sweep.py > $o/${r}_bf16_gemm_tile_sweep.txt 2>> $o/${r}_bench.err
python tools/gemm_8ph_probe.py > $o/${r}_bf16_gemm_8ph_time_model.txt 2>> $o/${r}_bench.err
# the N > 1 path, functionally: 2 and 8 RCCL ranks time-sharing the one GPU of the box (NOT a scaling number)
python bench.py --gpus 2 --share-gpu --steps 10 --warmup 2 --no-cpu-baseline > $o/${r}_bench_c2_train_2ranks_shared_gpu.json 2>> $o/${r}_bench.err
python bench.py --gpus 8 --share-gpu --steps 3 --warmup 1 --no-cpu-baseline > $o/${r}_bench_c2_train_8ranks_shared_gpu.json 2>> $o/${r}_bench.err
tools/prof_trace.sh ${r}_trace_c2_train --steps 5 --warmup 1 --no-cpu-baseline --no-split --no-legs
tools/prof_trace.sh ${r}_trace_c5_beam --mode beam --config c5 --steps 3 --warmup 1
tools/prof_trace.sh ${r}_trace_c1_beam1 --mode beam --config c1 --beam 1 --steps 20 --warmup 2
tools/prof_trace.sh ${r}_trace_c1_decode --mode decode --config c1 --steps 5 --warmup 1 --no-cpu-baseline
tools/prof_trace.sh ${r}_trace_c4_train_bf16 --config c4 --precision bf16 --steps 5 --warmup 1 --no-cpu-baseline --no-live-pmc
tools/prof_trace.sh ${r}_trace_eval_msvd --mode eval --eval-videos 128 --no-cpu-baseline
tools/prof_pmc.sh ${r}_pmc_c4_bf16 MfmaUtil python bench.py --config c4 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
    tools/prof_pmc.sh ${r}_pmc_train $c python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split --no-legs --no-live-pmc
done
tools/prof_pmc.sh ${r}_fetch_calibration FETCH_SIZE tools/bin/fetch_calib
# stamp every collected file with the HEAD it was measured at
for f in $o/${r}_*.csv; do sed -i "1i # git_head=$head sources_sha16=$src" $f; done
for f in $o/${r}_*.json; do
    [ -s $f ] && python - $f $head $src <<'PY'
import json, sys
p, head, src = sys.argv[1:4]
try:
    d = json.load(open(p))
except Exception:
    sys.exit(0)
if isinstance(d, dict):
    d["git_head"], d["sources_sha16"] = head, src
    json.dump(d, open(p, "w"))
PY
done
ls -la $o | grep ${r}_
