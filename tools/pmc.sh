# SQ / TCC counters of one kernel:  bash tools/pmc.sh NAME KERNEL WORK script args...
#   NAME: output tag; KERNEL: substring of the kernel name; WORK: divisor (e.g. rows x problems) for per-unit figures
# One rocprofv3 --pmc pass per counter group (never combined with tracing of other domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAME=$1; KERNEL=$2; WORK=$3; shift 3
SCRIPT=$R/$1; shift
OUT=$R/gpurun_out/pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $SCRIPT "$@" > $OUT/p$i.log 2>&1
  python3 $R/tools/pmc_rows.py $(find $OUT/p$i -name "*counter_collection.csv" | head -1) $KERNEL $WORK >> $OUT/${NAME}_counters.txt
  rm -rf $OUT/p$i
done
cat $OUT/${NAME}_counters.txt
