#!/bin/bash
# (runs on commit 695e981: the product no longer carries the kernel)
# ON THE GPU BOX: the materialising traversal with the one-lane-per-state kernel for small launches
# (clv_dna_quad_kernel; RDAMD_CLV_QUAD: 0 never, 2 = small launches only, 1 always) on an ABLATION
# build (the environment switch), two alternating rounds.  Shapes: c2, 125.phy's, c2 / 8's shard,
# c5's shard, c4's shard.   usage: profiles/clv_quad_ab.sh > gpurun_out/.../clv_quad_ab.txt
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
cp root_digger_amd/lib/librdamd_ablation.so root_digger_amd/lib/librdamd.so
for rep in 1 2; do
  for q in 0 2 1; do
    export RDAMD_CLV_QUAD=$q
    echo "== quad=$q"
    python profiles/clv_time.py 100 50000 4 20 4
    python profiles/clv_time.py 125 19436 4 20 4
    python profiles/clv_time.py 100 6250 4 20 4
    python profiles/clv_time.py 1000 50000 4 6 4
    python profiles/clv_time.py 500 62500 4 6 4
  done
done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
