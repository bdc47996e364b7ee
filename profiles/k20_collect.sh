#!/bin/bash
# The 20-state materialising traversal on the c3 shape, one box: kernel trace, memory-system
# counters, gap-free timings and the timing-only ablations (part of r2_collect.sh, here alone).
export TMPDIR=/tmp; R=/root/repo; OUT=$R/gpurun_out/r2; mkdir -p $OUT; cd /tmp
rm -rf $OUT/k20trace $OUT/k20pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k20trace -- python3 $R/profiles/clv_time.py 200 10000 4 20 20 > $OUT/k20trace.log 2>&1
bash $R/profiles/pmc_mem.sh r2/k20pmc traversal profiles/clv_time.py 200 10000 4 10 20 > $OUT/k20_pmc.txt 2>&1
cd $R; python3 profiles/clv_time.py 200 10000 4 20 20 > $OUT/k20_time.txt; python3 profiles/clv_time.py 200 100000 4 6 20 >> $OUT/k20_time.txt
python3 profiles/clv_time.py 200 12288 4 20 20 >> $OUT/k20_time.txt
python3 profiles/clv_time.py 100 50000 4 20 4 >> $OUT/k20_time.txt
VARS="0 1 16 8 4 29 0" bash profiles/k20_ab.sh > $OUT/k20_ablation.txt 2>&1
cat $OUT/k20_time.txt $OUT/k20_ablation.txt
