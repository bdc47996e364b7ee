#!/bin/bash
# the same A/B (profiles/clv_pieces_ab.sh) on the shapes that fill the device by themselves
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --no-shard-legs 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['clv_kernel']
e=os.environ.get
print('   pieces %-2s %-40s %8.2f us per traversal  %7.1f GB/s algorithmic' % (e('RDAMD_CLV_PIECES','-'), ' '.join(sys.argv[1:]), 1e3 * k['avg_launch_ms'], k['achieved']))" "$@"
}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for cfg in "--config c5 --steps 2 --warmup 1" "--config c4 --steps 1 --warmup 1" "--config c4 --sites 62500 --steps 2 --warmup 1" "--config c2 --sites 6250 --steps 5 --warmup 2"; do
  for pcs in ${PIECES:-0 4 8 16 32}; do RDAMD_CLV_PIECES=$pcs one $cfg; done
done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
