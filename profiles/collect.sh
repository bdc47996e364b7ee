#!/bin/bash
# Round profile collection (run on the GPU box through gpurun):
#   bash profiles/collect.sh r1
# 1. rocprofv3 --kernel-trace --stats of the default bench command
# 2. separate PMC passes (FETCH_SIZE / WRITE_SIZE need different TCC slots)
# Outputs go to gpurun_out/<tag>/ ; profiles/summarize.py turns them into
# profiles/<tag>_*.csv + profiles/<tag>_summary.json (committed).
TAG=${1:-r1}
export TMPDIR=/tmp; R=/root/repo; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/sq3 -- $BENCH > $OUT/sq3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/mem1 -- $BENCH > $OUT/mem1.log 2>&1
$BENCH > $OUT/bench_plain.log 2>&1
grep -h '"metric"' $OUT/bench_plain.log | cut -c1-200
ls $OUT
