#!/bin/bash
# The A/B runs of the kernels under csrc/experimental/ and of the compile-time probe builds, in one gpurun call.  Before the call (here;
# tools/_var travels with the snapshot):   tools/build_all_variants.sh
#   gpurun --timeout 3000 -- tools/gpu_session_r06.sh ab      -> gpurun_out/next_session_report.txt
# Decides (promote into the product sources, or delete -- VERDICT r05 item 2; bars in parentheses):
#   dpp      -DSTATTN_DPP_REDUCE=1          wave reductions on DPP lane moves instead of ds_bpermute chains (whole suite green + no leg slower)
#   exp      STATTN_BWD2=1|2                spatial_bwd2_kernel            (spatial_bwd_kernel<8> 48.6 us -> <= 43 us)
#   exp      STATTN_BF16_V2=1               spatial_bf16v2 / spatial_bwd_bf16v2 (68 -> <= 55 us, 100 -> <= 80 us with no scratch)
#   exp      STATTN_SHARED_COLS=1           spatial_shared_cols_kernel     (eval shared-slab launch 92 -> <= 70 us)
#   wps2 / fwdsched                         one-line variants of the shipped bf16 attention kernels
#   pnv2     -DSTATTN_PN_V2=1               row-panel kernels with the prologue rewritten against its ISA (panel.hip): no exposed early
#                                           round trip, half the scalar waits (panel_kernel<4,1,512,4> 11.6 -> <= 10 us, lstm 12.5 -> <= 11 us,
#                                           whole suite green, results bit-identical to the product's)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=gpurun_out; mkdir -p $o
STAGES=${STAGES:-"6 0 1 2 3 4 5"}
rep=$o/next_session_report.txt; : >> $rep
stage() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
say() { echo "$@" | tee -a $rep; }
line() { python - "$1" <<'PY' 2>/dev/null
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  (no bench line: %s)" % e); sys.exit(0)
k = d.get("kernels", {})
def ms(n): return "%.1f" % (k[n]["ms_per_launch"] * 1e3) if n in k else "-"
print("  %.3f ms/step  %.1f k row-steps/s | spatial %s us  bwd_spatial %s us  temporal %s us  ctxgrad %s us | eval videos/s %s  us/word %s" % (
    d.get("ms_per_step", 0), d.get("value", 0) / 1e3, ms("spatial"), ms("bwd_spatial"), ms("temporal"), ms("bwd_ctxgrad"), d.get("videos_per_s"), d.get("us_per_word")))
PY
}
bench() { tag=$1; shift; timeout 400 "$@" > $o/ns_$tag.json 2> $o/ns_$tag.err; say "$tag:"; line $o/ns_$tag.json | tee -a $rep; }
V=tools/with_variant.sh
T="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split --no-legs --no-live-pmc"
C4="python bench.py --config c4 --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-legs --no-live-pmc"

if stage 6; then
say "== 6. row-panel prologue V2 (variant pnv2): headline step, decode legs, then the whole GPU suite under it"
bench c2_product_a $T
bench c2_pnv2 $V pnv2 $T
bench c2_product_b $T
bench c2_pnv2_b $V pnv2 $T
for m in "--mode decode --config c1 --steps 5 --warmup 1" "--mode beam --config c1 --beam 5 --steps 50 --warmup 3" "--mode beam --config c5 --steps 5 --warmup 1"; do
    t=$(echo $m | tr -d ' -'); bench ${t}_product python bench.py $m --no-cpu-baseline; bench ${t}_pnv2 $V pnv2 python bench.py $m --no-cpu-baseline
done
$V pnv2 timeout 2400 python -m pytest tests -m gpu -x -q > $o/ns_tests_pnv2.log 2>&1; say "pnv2 whole suite: $(tail -1 $o/ns_tests_pnv2.log)"
fi
if stage 0; then
say "== 0. wave_sum / wave_max (shuffle and DPP forms) against a host loop"
tools/bin/dpp_check0 2>&1 | tail -1 | tee -a $rep
tools/bin/dpp_check1 2>&1 | tail -1 | tee -a $rep
fi
if stage 1; then
say "== 1. configs[1] train step: product build, the DPP build, spatial_bwd2 under both"
bench c2_product $T
bench c2_dpp $V dpp $T
STATTN_BWD2=1 bench c2_bwd2_4wg $V exp $T
STATTN_BWD2=2 bench c2_bwd2_3wg $V exp $T
STATTN_BWD2=1 bench c2_dpp_bwd2_4wg $V expdpp $T
STATTN_BWD2=2 bench c2_dpp_bwd2_3wg $V expdpp $T
bench c2_product_again $T
fi
if stage 2; then
say "== 2. parity: whole GPU suite under the DPP build; backward + property tests under spatial_bwd2 (both register budgets, both reductions)"
$V dpp timeout 2400 python -m pytest tests -m gpu -x -q > $o/ns_tests_dpp.log 2>&1; tail -2 $o/ns_tests_dpp.log | tee -a $rep
for v in 1 2; do
    STATTN_BWD2=$v $V exp timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_properties.py -m gpu -x -q > $o/ns_tests_bwd2_$v.log 2>&1; say "STATTN_BWD2=$v: $(tail -1 $o/ns_tests_bwd2_$v.log)"
    STATTN_BWD2=$v $V expdpp timeout 1200 python -m pytest tests/test_gpu_backward.py -m gpu -x -q > $o/ns_tests_dpp_bwd2_$v.log 2>&1; say "STATTN_BWD2=$v + DPP: $(tail -1 $o/ns_tests_dpp_bwd2_$v.log)"
done
fi
if stage 3; then
say "== 3. configs[3] bf16 step and the other legs under the variants"
bench c4_bf16_product $C4
bench c4_bf16_dpp $V dpp $C4
bench c4_bf16_wps2 $V wps2 $C4
bench c4_bf16_fwdsched $V fwdsched $C4
STATTN_BF16_V2=1 bench c4_bf16_v2 $V exp $C4
STATTN_BF16_V2=1 bench c4_bf16_v2_dpp $V expdpp $C4
STATTN_BF16_V2=1 $V exp timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $o/ns_tests_bf16v2.log 2>&1; say "STATTN_BF16_V2=1 test_gpu_bf16: $(tail -1 $o/ns_tests_bf16v2.log)"
STATTN_BF16_V2=1 $V exp timeout 900 python tools/fuzz_parity.py 45 4242 > $o/ns_fuzz_bf16v2.log 2>&1; say "STATTN_BF16_V2=1 fuzz (every third case a bf16 handle): $(tail -1 $o/ns_fuzz_bf16v2.log)"
bench c5_product python bench.py --mode beam --config c5 --steps 5 --warmup 1 --no-cpu-baseline
bench c5_dpp $V dpp python bench.py --mode beam --config c5 --steps 5 --warmup 1 --no-cpu-baseline
bench eval_product python bench.py --mode eval --no-cpu-baseline
bench eval_dpp $V dpp python bench.py --mode eval --no-cpu-baseline
fi
if stage 4; then
say "== 4. column-per-lane shared attention (K <= 8 beams): parity, then the evaluation workload"
STATTN_SHARED_COLS=1 $V exp timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $o/ns_tests_cols.log 2>&1; say "STATTN_SHARED_COLS=1: $(tail -1 $o/ns_tests_cols.log)"
STATTN_SHARED_COLS=1 $V expdpp timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/ns_tests_cols_dpp.log 2>&1; say "STATTN_SHARED_COLS=1 + DPP: $(tail -1 $o/ns_tests_cols_dpp.log)"
STATTN_SHARED_COLS=1 bench eval_cols $V exp python bench.py --mode eval --no-cpu-baseline
STATTN_SHARED_COLS=1 bench eval_cols_dpp $V expdpp python bench.py --mode eval --no-cpu-baseline
STATTN_SHARED_COLS=1 STATTN_SHARED_MIN=100 bench eval_cols_dpp_min100 $V expdpp python bench.py --mode eval --no-cpu-baseline
fi
if stage 5; then
say "== 5. evaluation chunks beyond 255 rows on the row-panel kernels (tools build, STATTN_PANEL_MAX_ROWS=512): parity of a 102-video call, then the leg"
for ch in 51 64 80 102; do
    bench eval_chunk${ch}_rule256 $V tools python bench.py --mode eval --eval-chunk $ch --no-cpu-baseline
    STATTN_PANEL_MAX_ROWS=512 bench eval_chunk${ch}_rule512 $V tools python bench.py --mode eval --eval-chunk $ch --no-cpu-baseline
done
STATTN_PANEL_MAX_ROWS=512 STATTN_EVAL_SHAPE_VIDEOS=102 $V tools timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msvd_eval_shape or batched_beam_search or c5_" > $o/ns_tests_rows512.log 2>&1; say "STATTN_PANEL_MAX_ROWS=512 beam tests (102 videos at the evaluation shape): $(tail -1 $o/ns_tests_rows512.log)"
fi
say "== done ($STAGES)"
