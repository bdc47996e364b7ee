# fused 4-state evaluator: bench lines for configs/variants:
#   CONFIGS="c2 c4" NSS="0 1 2" bash profiles/fd_ab.sh     (NS 0 = the library's own choice)
# (the timing-only variants live in the ablation build only: csrc `make ablation`)
make -s -C root_digger_amd/csrc ablation >/dev/null; ABL=$PWD/root_digger_amd/lib/librdamd_ablation.so
for c in ${CONFIGS:-c2}; do for ns in ${NSS:-0}; do
RDAMD_FUSED_NS=$ns python profiles/with_ablation.py $ABL bench.py --config $c --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$c ns=$ns evals/s %.0f  kernel ms %.3f  frac %.3f  depth %d' % (d['value'], r['avg_launch_ms'], r['frac'], r['stack_depth']))"; done; done
