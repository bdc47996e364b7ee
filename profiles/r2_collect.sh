#!/bin/bash
# Round-2 evidence, one box: kernel trace + PMC passes of the default bench command
# (profiles/collect.sh), the bench lines of every BASELINE config, and the 20-state
# kernels' trace + memory-system counters.  Outputs under gpurun_out/r2/.
export TMPDIR=/tmp; R=/root/repo; OUT=$R/gpurun_out/r2; mkdir -p $OUT
bash $R/profiles/collect.sh r2 > $OUT/collect.log 2>&1
cd $R
python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 bench.py --config c3 --steps 5 --warmup 1 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3trace -- python3 $R/bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/c3trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k20trace -- python3 $R/profiles/clv_time.py 200 10000 4 20 20 > $OUT/k20trace.log 2>&1
bash $R/profiles/pmc_mem.sh r2/k20pmc traversal profiles/clv_time.py 200 10000 4 10 20 > $OUT/k20_pmc.txt 2>&1
bash $R/profiles/pmc_mem.sh r2/f20pmc fused20_eval bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/f20_pmc.txt 2>&1
cd $R; python3 profiles/clv_time.py 200 10000 4 20 20 > $OUT/k20_time.txt; python3 profiles/clv_time.py 200 100000 4 6 20 >> $OUT/k20_time.txt
python3 profiles/clv_time.py 100 50000 4 20 4 >> $OUT/k20_time.txt
ls $OUT
