#!/bin/bash
# usage: bash profiles/pmc_mem.sh <tag> <kernel name filter> <script.py> [args]
#   L1 (TCP/TA), L2 (TCC), fabric and SQ counters of the kernels whose name contains the filter,
#   one rocprofv3 --pmc pass per counter set (no tracing options beside them)
#   e.g.  bash profiles/pmc_mem.sh k20 traversal profiles/clv_time.py 200 10000 4 10 20
#         bash profiles/pmc_mem.sh f20 fused20_eval bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline
TAG=$1; FILTER=$2; SCRIPT=$3; shift 3
ARGS="$@"
export TMPDIR=/tmp; R=/root/repo; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/$SCRIPT $ARGS > $OUT/p$i.log 2>&1
done
python3 - $OUT $FILTER <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-64:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if sys.argv[2] in k:
        print(k)
        for c, vals in sorted(v.items()):
            print("  %-34s %16.0f (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
