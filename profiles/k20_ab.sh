# A/B of the 20-state materialising traversal kernel's timing-only variants (RDAMD_K20_VAR)
# (the timing-only variants live in the ablation build only: csrc `make ablation`)
make -s -C root_digger_amd/csrc ablation >/dev/null; ABL="profiles/with_ablation.py $PWD/root_digger_amd/lib/librdamd_ablation.so"
[ -n "$PARITY" ] && timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "protein or mfma or 20 or arbitrary or generic" 2>&1 | tail -3
for v in ${VARS:-0 1 5 13 17 29 0}; do echo VAR=$v; RDAMD_K20_VAR=$v python $ABL profiles/clv_time.py 200 10000 4 20 20; [ -n "$BIG" ] && RDAMD_K20_VAR=$v python $ABL profiles/clv_time.py 200 100000 4 6 20; done
