#!/bin/bash
# The shapes ONE GPU sees under BASELINE's sharding: c4 / 8 = 500 taxa x 62 500
# sites, c5 as 4 candidate groups x 2 site shards = 1000 taxa x 50 000 sites.  Bench lines + the
# memory-system counters of the fused evaluator (L2 hit / miss).  gpurun -- 'bash profiles/r4_shard.sh'
O=${OUT:-gpurun_out/r4_shard}; mkdir -p $O
B="python3 bench.py --allow-stale-profile --no-cpu-baseline --sustain-seconds 0"
$B --config c4 --sites 62500 --steps 4 --warmup 1 > $O/c4_shard_bench.json 2> $O/c4_shard.err
$B --config c5 --sites 50000 --steps 4 --warmup 1 > $O/c5_shard_bench.json 2> $O/c5_shard.err
for c in c4 c5; do python3 -c "
import json
d=json.load(open('$O/${c}_shard_bench.json')); r=d['roofline']
print('$c shard: %9.1f evals/s  kernel %8.3f ms  frac %.4f  steps %.1f of %d  depth %d' % (d['value'], r['avg_launch_ms'], r['frac'], r['schedule']['steps_per_evaluation'], r['schedule']['operations_per_evaluation'], r['stack_depth']))"; done | tee $O/summary.txt
if [ -z "$NOPMC" ]; then
export TMPDIR=/tmp; cd /tmp
for c in c4:62500 c5:50000; do
  cfg=${c%%:*}; sites=${c##*:}
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$O/${cfg}_p$i -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --sites $sites --steps 2 --warmup 1 --allow-stale-profile --no-cpu-baseline --sustain-seconds 0 > $GRAFT_REPO_ROOT/$O/${cfg}_p$i.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 - $O <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, collections
for cfg in ("c4", "c5"):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/%s_p*/*/*counter_collection.csv" % (sys.argv[1], cfg)):
        for r in csv.DictReader(open(f)):
            if "fused_dna_eval_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    if not m: continue
    print(cfg, "shard, fused_dna_eval_kernel, per launch:")
    for k in sorted(m): print("   %-28s %16.0f" % (k, m[k]))
    if m.get("TCC_REQ_sum"): print("   L2 miss ratio %.3f" % (m["TCC_MISS_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
    rd = m.get("TCC_EA0_RDREQ_sum", 0); rd32 = m.get("TCC_EA0_RDREQ_32B_sum", 0)
    print("   bytes read beyond L2  %.2f GB, written %.2f GB" % ((rd32 * 32 + (rd - rd32) * 64) / 1e9,
          (m.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (m.get("TCC_EA0_WRREQ_sum", 0) - m.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32) / 1e9))
PY
find $O -name '*.csv' -size +2M -delete; rm -rf $O/*_p[0-9]
fi
cat $O/summary.txt
