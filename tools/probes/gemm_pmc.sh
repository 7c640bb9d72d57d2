#!/bin/bash
# PMC passes (each its own rocprofv3 run, counters only + kernel trace) over ONE fp32 GEMM shape: where the matrix pipe's idle cycles go
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-gemm}; shift
o=$ROOT/gpurun_out; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"; do
    i=$((i+1))
    d=$o/pmc_${tag}_$i; rm -rf $d
    ( cd $ROOT && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o $tag -- "$@" ) > $o/${tag}_pmc$i.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $ROOT/tools/pmc_summary.py "$f" | grep -i "gemm\|^kernel" > $o/${tag}_pmc$i.csv
    rm -rf $d
done
cat $o/${tag}_pmc*.csv
