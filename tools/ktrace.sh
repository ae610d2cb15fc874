# per-dispatch kernel trace of one command (on the GPU box, from the repo root):
#   bash tools/ktrace.sh NAME script.py [args...]  ->  gpurun_out/r04/NAME_kernel_trace.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
name=$1; shift; script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_$name
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$name -- python3 $script "$@" > $O/$name.log 2>&1
cp $(find $O/trace_$name -name "*kernel_trace.csv" | head -1) $O/${name}_kernel_trace.csv
rm -rf $O/trace_$name
grep -v "^[EWI]2026\|amdgpu.ids" $O/$name.log | tail -3
