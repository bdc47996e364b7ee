#!/bin/bash
# the 20-state traversal kernel's cut (kernels_clv_mfma.hip clv_k20_traversal_cut; round 3: 8 pieces, one level) with
# more pieces and levels, one box, ablation library (knobs: profiles/clv_pieces_ab.sh)
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['clv_kernel']
e=os.environ.get
print('   pieces %-2s ops %-2s min %-2s %-40s %8.2f us per traversal (%d launches)  frac %.4f' % (e('RDAMD_CLV_PIECES','-'), e('RDAMD_CLV_PIECE_OPS','-'), e('RDAMD_CLV_MIN_SPLIT','-'), ' '.join(sys.argv[1:]), 1e3 * k['avg_launch_ms'], k.get('kernel_launches_per_traversal', 1), k['frac']))" "$@"
}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for cfg in "--config c3 --steps 2 --warmup 1" ${MORE_CFG:+"$MORE_CFG"}; do
  one $cfg
  RDAMD_CLV_PIECES=0 one $cfg
  for pcs in ${PIECES:-8 16 32}; do for po in ${PIECE_OPS:-12 24}; do for ms in ${MIN_SPLIT:-12}; do
    RDAMD_CLV_PIECES=$pcs RDAMD_CLV_PIECE_OPS=$po RDAMD_CLV_MIN_SPLIT=$ms one $cfg
  done; done; done
  one $cfg
done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
