#!/bin/bash
# Round-6 evidence on ONE box (gpurun -- 'bash profiles/r6_collect.sh'):
#  1. collect.sh r6: rocprofv3 --kernel-trace --stats and the separate PMC passes of the default bench command (c2);
#  2. the same for `bench.py --config c3` (the 20-state evaluator: matrix-pipe and address-unit counters) -> r6/c3_*;
#     profiles/summarize.py r6 (run in the container afterwards, on the SAME sources) turns both into
#     profiles/r6_kernel_stats.csv, r6_c3_kernel_stats.csv, r6_summary.json (digest-bound);
#  3. the bench line of every BASELINE config, 125.phy and the shard shapes -> gpurun_out/r6/<name>_bench.json.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
bash profiles/collect.sh r6 > $O/collect.log 2>&1
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
B3="python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0"
( cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/c3_trace -- $B3 > $R/$O/c3_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/c3_fetch -- $B3 > $R/$O/c3_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/c3_write -- $B3 > $R/$O/c3_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/$O/c3_sq1 -- $B3 > $R/$O/c3_sq1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$O/c3_sq2 -- $B3 > $R/$O/c3_sq2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$O/c3_mem1 -- $B3 > $R/$O/c3_mem1.log 2>&1 )
ls $O | head -40
