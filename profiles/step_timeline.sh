#!/bin/bash
# ON THE GPU BOX: every profiles/tmp_libs/stamps_<mask>.so (profiles/step_timeline_build.sh) takes the
# product library's place in turn; profiles/step_timeline.py prints cycles per phase of RDAMD_STEP
# for the shapes given.  The product library's own bench line (kernel ms) goes first and last, so
# that the instrument's cost is on the same page.
# usage: profiles/step_timeline.sh <outdir> "<masks>" "<shapes>"
O=${1:-gpurun_out/step_timeline}; MASKS=${2:-31 1 3 5 9 17}; SHAPES=${3:-c2 c4s c5s}
mkdir -p $O
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
prod() {
  for c in "--steps 10 --warmup 2" "--config c4 --sites 62500 --steps 3 --warmup 1" "--config c5 --sites 50000 --steps 3 --warmup 1"; do
    python bench.py $c --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('product  %-40s %9.1f evals/s  kernel %8.4f ms  frac %.4f' % (sys.argv[1], d['value'], r['avg_launch_ms'], r['frac']))" "$c"
  done
}
prod | tee $O/product_before.txt
for m in $MASKS; do
  cp profiles/tmp_libs/stamps_$m.so root_digger_amd/lib/librdamd.so
  for s in $SHAPES; do
    timeout 600 python profiles/step_timeline.py --config $s --label mask$m --json $O/${s}_mask$m.json 2>&1 | tee -a $O/timeline.txt
  done
done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
prod | tee $O/product_after.txt
