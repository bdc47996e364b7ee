#!/bin/bash
# one wave per rate category (FusedArgs::rates_across_waves: the code arena is fetched once per workgroup instead of once per
# rate pass) on the shard shapes, where the product does not take it (arena under 512 MB); ablation library, RDAMD_FUSED_RW
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   RW %-2s %-44s %9.1f evals/s  kernel %8.4f ms  frac %.4f' % (os.environ.get('RDAMD_FUSED_RW','-'), ' '.join(sys.argv[1:]), d['value'], r['avg_launch_ms'], r['frac']))" "$@"
}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for rep in 1 2; do for cfg in "--config c4 --sites 62500 --steps 4 --warmup 1" "--config c5 --sites 50000 --steps 4 --warmup 1" "--config c5 --steps 3 --warmup 1"; do
  RDAMD_FUSED_RW=0 one $cfg; RDAMD_FUSED_RW=1 one $cfg
done; done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
