#!/bin/bash
# the library's own cut (clv_traversal_pieces, kernels_clv.hip) against the whole list, alternating, one box
# (ablation library: RDAMD_CLV_PIECES=0 switches the cut off)
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['clv_kernel']
print('   %-10s %-44s %8.2f us per traversal (%d launches)  %7.1f GB/s algorithmic  frac %.4f' % ('whole list' if os.environ.get('RDAMD_CLV_PIECES') else 'cut', ' '.join(sys.argv[1:]), 1e3 * k['avg_launch_ms'], k.get('kernel_launches_per_traversal', 1), k['achieved'], k['frac']))" "$@"
}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for rep in 1 2; do
  for cfg in "--steps 5 --warmup 2" "--config d125 --steps 5 --warmup 2" "--config c2 --sites 6250 --steps 5 --warmup 2" "--config c5 --sites 50000 --steps 2 --warmup 1" "--config c4 --sites 62500 --steps 2 --warmup 1" "--config c5 --steps 2 --warmup 1"; do
    RDAMD_CLV_PIECES=0 one $cfg
    one $cfg
  done
done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
