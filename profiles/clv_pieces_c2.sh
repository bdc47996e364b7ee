#!/bin/bash
# c2 only: pieces per launch x the size a piece keeps (ablation library; see clv_pieces_ab.sh)
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['clv_kernel']
e=os.environ.get
print('   pieces %-2s ops %-2s min %-2s %8.2f us per traversal (%d launches)' % (e('RDAMD_CLV_PIECES','-'), e('RDAMD_CLV_PIECE_OPS','-'), e('RDAMD_CLV_MIN_SPLIT','-'), 1e3 * k['avg_launch_ms'], k.get('kernel_launches_per_traversal', 1)))" "$@"
}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for rep in 1 2; do
  one --steps 5 --warmup 2
  for pcs in ${PIECES:-4 6 8 10 12}; do for po in ${PIECE_OPS:-8 16 24}; do
    RDAMD_CLV_PIECES=$pcs RDAMD_CLV_PIECE_OPS=$po RDAMD_CLV_MIN_SPLIT=${MIN_SPLIT:-16} one --steps 5 --warmup 2
  done; done
done
RDAMD_CLV_PIECES=10 RDAMD_CLV_PIECE_OPS=16 RDAMD_CLV_DEBUG=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>&1 >/dev/null | grep "clv pieces" | head -4
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
