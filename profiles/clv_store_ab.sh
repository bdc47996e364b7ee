#!/bin/bash
# A/B on ONE box: clv_dna_traversal_kernel's stores as the lane's own two 16-byte halves (tmp_libs/plain.so: an ablation
# build with -DRDAMD_ABL_PLAIN_STORES) against quad-swizzled stores, 64 contiguous bytes per quad and instruction
# (tmp_libs/swizzle.so = the product library).  Figures: the materialising leg of the bench line (clv_kernel).
one() {
  python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['clv_kernel']
print('   %-34s %s  %8.2f us per traversal  %7.1f GB/s algorithmic  frac %.4f' % (' '.join(sys.argv[1:]), k['kernel'], 1e3 * k['avg_launch_ms'], k['achieved'], k['frac']))" "$@"
}
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
for rep in 1 2; do for l in ${LIBS:-plain swizzle}; do cp profiles/tmp_libs/$l.so root_digger_amd/lib/librdamd.so; echo "== $l"; one --steps 5 --warmup 2; one --config d125 --steps 5 --warmup 2; one --config c5 --sites 50000 --steps 2 --warmup 1; done; done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
