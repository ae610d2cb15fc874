# kernel-trace summary of one command (run on the GPU box through gpurun, from the repo root):
#   bash tools/prof.sh NAME script.py [args...]   ->  gpurun_out/r04/r04_NAME_kernel_stats.csv (+ NAME.log)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
name=$1; shift; script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$name
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $script "$@" > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/r04_${name}_kernel_stats.csv
rm -rf $O/prof_$name
grep -v "^[EWI]2026\|amdgpu.ids" $O/$name.log | tail -5
