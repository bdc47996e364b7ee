# A/B of the 20-state materialising traversal kernel's timing-only variants (RDAMD_K20_VAR)

for v in ${VARS:-0 1 2 3 0}; do echo VAR=$v; RDAMD_K20_VAR=$v python profiles/clv_time.py 200 10000 4 20 20; done
python profiles/clv_time.py 200 100000 4 6 20
