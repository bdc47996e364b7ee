cd $GRAFT_REPO_ROOT
make -s -C root_digger_amd/csrc ablation > /dev/null 2>&1
ABL=$PWD/root_digger_amd/lib/librdamd_ablation.so
for cfg in "c5 50000" "c4 62500"; do set -- $cfg
for ns in 2 1; do
RDAMD_FUSED_NS=$ns python3 profiles/with_ablation.py $ABL bench.py --config $1 --sites $2 --steps 3 --warmup 1 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 shard ns=$ns %9.1f evals/s kernel %.3f ms frac %.3f' % (d['value'], r['avg_launch_ms'], r['frac']))"
done; done
