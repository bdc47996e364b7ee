#!/bin/bash
# Timing-only A/B: what is the rescale test of the fused 4-state evaluator
# worth?  The ablation library is built three times on the box: as shipped, without the test on
# the running-CLV x tip steps, without any test (results are garbage wherever a rescale is due;
# on c2 none ever is).  Usage: gpurun -- 'bash profiles/rescale_ab.sh'
O=gpurun_out/rescale_ab; mkdir -p $O
for v in 0 1 2; do
  X=""; [ $v != 0 ] && X="-DRDAMD_ABL_NOCHECK=$v"
  rm -f root_digger_amd/csrc/build/abl/kernels_fused.hip.o
  make -s -C root_digger_amd/csrc ablation ABL_EXTRA="$X" > $O/build_$v.log 2>&1 || { tail -5 $O/build_$v.log; exit 1; }
  for c in ${CONFIGS:-c2 c5}; do
    st="--steps 20 --warmup 3"; [ $c = c5 ] && st="--steps 3 --warmup 1"
    RDAMD_BENCH_TIMING_ONLY=1 python3 profiles/with_ablation.py $PWD/root_digger_amd/lib/librdamd_ablation.so bench.py --config $c $st \
      --allow-stale-profile --no-cpu-baseline --sustain-seconds 0 2> $O/${c}_$v.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-3s test variant $v: %9.1f evaluations/s  kernel %.3f ms' % ('$c', d['value'], r['avg_launch_ms']))" | tee -a $O/result.txt
  done
done
rm -f root_digger_amd/csrc/build/abl/kernels_fused.hip.o
