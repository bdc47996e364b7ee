#!/bin/bash
# A/B on ONE box: libraries profiles/tmp_libs/<name>.so (ablation builds of csrc/Makefile with
# different ABL_EXTRA) take the product library's place in turn, two alternating rounds, four
# workloads each.  usage: profiles/lib_ab.sh base prio1 prio4 ...
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   %-28s %9.1f evals/s  kernel %8.4f ms  frac %.4f' % (' '.join(sys.argv[1:]), d['value'], r['avg_launch_ms'], r['frac']))" "$@"
}
CMD=${LIB_AB_CMD:-'one --steps 20 --warmup 3; one --config c4 --sites 62500 --steps 4 --warmup 1; one --config c5 --sites 50000 --steps 4 --warmup 1; one --config d125 --steps 20 --warmup 3'}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
for rep in 1 2; do for l in "$@"; do cp profiles/tmp_libs/$l.so root_digger_amd/lib/librdamd.so; echo "== $l"; eval "$CMD"; done; done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
