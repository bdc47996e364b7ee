#!/bin/bash
# clv_dna_traversal_kernel with its operation list cut into independent pieces that run side by side,
# level by level (kernels_clv.hip, k20_split.hpp list_levels, partition.hip), one box.  Needs the
# ablation library (make -C root_digger_amd/csrc ablation) for the knobs:
#   RDAMD_CLV_PIECES       most pieces per launch (0: the list as it is)
#   RDAMD_CLV_PIECE_OPS    a piece of at most this many operations is not cut further
#   RDAMD_CLV_MIN_SPLIT    a (remaining) list shorter than this runs as one piece
#   RDAMD_CLV_PIECE_SLOTS  LDS parking slots of launches with several pieces (default: chosen by read-backs)
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['clv_kernel']
e=os.environ.get
print('   pieces %-2s ops %-2s min %-2s slots %-2s %-40s %8.2f us per traversal  %7.1f GB/s algorithmic' % (e('RDAMD_CLV_PIECES','-'), e('RDAMD_CLV_PIECE_OPS','-'), e('RDAMD_CLV_MIN_SPLIT','-'), e('RDAMD_CLV_PIECE_SLOTS','-'), ' '.join(sys.argv[1:]), 1e3 * k['avg_launch_ms'], k['achieved']))" "$@"
}
dbg() { RDAMD_CLV_DEBUG=1 python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>&1 >/dev/null | grep "clv pieces" | head -${DBG_LINES:-6}; }
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for rep in $(seq ${REPS:-2}); do
  for cfg in "--steps 5 --warmup 2" "--config d125 --steps 5 --warmup 2" "--config c5 --sites 50000 --steps 2 --warmup 1" ${MORE_CFG:+"$MORE_CFG"}; do
    RDAMD_CLV_PIECES=0 one $cfg
    for pcs in ${PIECES:-8 16 32}; do for po in ${PIECE_OPS:-8 12}; do for ms in ${MIN_SPLIT:-8 16}; do
      export RDAMD_CLV_PIECES=$pcs RDAMD_CLV_PIECE_OPS=$po RDAMD_CLV_MIN_SPLIT=$ms
      one $cfg
      for sl in $SLOTS; do RDAMD_CLV_PIECE_SLOTS=$sl one $cfg; done
      [ -n "$DEBUG" ] && dbg $cfg
      unset RDAMD_CLV_PIECES RDAMD_CLV_PIECE_OPS RDAMD_CLV_MIN_SPLIT
    done; done; done
    one $cfg   # the library's own choice
  done
done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
