# usage: bash profiles/run_pmc.sh <tag>; SQ counter passes over bench.py (diagnostic)
TAG=$1; shift
export TMPDIR=/tmp; R=/root/repo; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
for e in "$@"; do export "$e"; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM --output-format csv -d $OUT/pmc2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU --output-format csv -d $OUT/pmc3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc3.log 2>&1
grep -h '"metric"' $OUT/pmc1.log | cut -c1-120
