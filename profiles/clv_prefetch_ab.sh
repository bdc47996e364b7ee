#!/bin/bash
# (runs on commit 695e981: the product no longer carries the switch)
# ON THE GPU BOX: the materialising traversal with / without "a marked memory operand is requested one
# operation ahead, in front of the stores" (kernels_clv.hip PF, round 6) on an ABLATION build
# (RDAMD_CLV_AHEAD: 0 never, 2 = the library's rule -- launches where >= 40 % of the operations have
# such an operand, i.e. the joining launches of a cut list --, 1 every launch), two alternating rounds.
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for rep in 1 2; do
  for a in 0 2 1; do
    export RDAMD_CLV_AHEAD=$a
    echo "== look-ahead=$a"
    python profiles/clv_time.py 100 50000 4 20 4
    python profiles/clv_time.py 125 19436 4 20 4
    python profiles/clv_time.py 100 6250 4 20 4
    python profiles/clv_time.py 1000 50000 4 6 4
    python profiles/clv_time.py 500 62500 4 6 4
  done
done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
