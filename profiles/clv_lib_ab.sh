#!/bin/bash
# the materialising traversal (clv_kernel leg of the bench line) under libraries profiles/tmp_libs/<name>.so, alternating,
# one box; CHECK=1: the parity tests under each library first.   usage: profiles/clv_lib_ab.sh base variant ...
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['clv_kernel']
print('   %-46s %8.2f us per traversal (%d launches)' % (' '.join(sys.argv[1:]), 1e3 * k['avg_launch_ms'], k.get('kernel_launches_per_traversal', 1)))" "$@"
}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
for rep in 1 2; do for l in "$@"; do
  cp profiles/tmp_libs/$l.so root_digger_amd/lib/librdamd.so; echo "== $l"
  if [ -n "$CHECK" ] && [ $rep = 1 ]; then timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse.py -q --timeout=120 2>&1 | tail -2; fi
  one --steps 5 --warmup 2; one --config d125 --steps 5 --warmup 2; one --config c2 --sites 6250 --steps 5 --warmup 2
  one --config c5 --sites 50000 --steps 2 --warmup 1; one --config c4 --sites 62500 --steps 2 --warmup 1
done; done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
