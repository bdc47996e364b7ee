cd $GRAFT_REPO_ROOT
make -s -C root_digger_amd/csrc ablation > /dev/null 2>&1
ABL=$PWD/root_digger_amd/lib/librdamd_ablation.so
for cfg in "c2 50000 20 3" "c5 50000 3 1" "c4 62500 3 1" "d125 0 20 3"; do set -- $cfg
for sh in 0 1; do
S=""; [ $2 != 0 ] && S="--sites $2"
RDAMD_FUSED_SHARE=$sh timeout 600 python3 profiles/with_ablation.py $ABL bench.py --config $1 $S --steps $3 --warmup $4 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 share=$sh %9.1f evals/s kernel %.3f ms frac %.3f check %.6f' % (d['value'], r['avg_launch_ms'], r['frac'], d['lnl_check']))"
done; done
