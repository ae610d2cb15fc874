# kernel-time profile of one command on the GPU box:  bash tools/prof.sh NAME tools/configs.py cfg4s
# (rocprofv3 --kernel-trace --stats; the per-kernel summary lands in gpurun_out/prof_NAME/NAME_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAME=$1; shift
OUT=$R/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
SCRIPT=$R/$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $SCRIPT "$@" > $OUT/log.txt 2>&1
echo "rc=$?"
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
cp $f $OUT/${NAME}_kernel_stats.csv
[ -n "$PROF_TRACE" ] && cp $(find $OUT/raw -name "*kernel_trace.csv" | head -1) $OUT/${NAME}_kernel_trace.csv
rm -rf $OUT/raw
head -${PROF_LINES:-16} $OUT/${NAME}_kernel_stats.csv | cut -c1-150
tail -4 $OUT/log.txt | cut -c1-400
