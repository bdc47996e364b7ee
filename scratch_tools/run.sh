export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/c3prof; rm -rf $OUT; mkdir -p $OUT; cd /tmp
python3 $R/bench.py --config c3 --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc2 -- python3 $R/bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs head -6
