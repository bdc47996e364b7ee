#!/bin/bash
# usage: bash profiles/k20_pmc.sh <tag> [n S R reps K]  -- L1/L2/fabric counters of the materialising traversal kernel
TAG=$1; shift
ARGS=${@:-200 10000 4 10 20}
export TMPDIR=/tmp; R=/root/repo; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/profiles/clv_time.py $ARGS > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "traversal" in k:
        print(k)
        for c, vals in sorted(v.items()):
            print("  %-34s %16.0f (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
