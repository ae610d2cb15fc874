# end-of-round profile set (run from the repo root on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
echo "== bench" ; timeout -k 10 900 python3 $R/bench.py > $O/r02_bench.json 2> $O/bench.err; echo rc=$?
echo "== rocprof bench"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --no-cpu-baseline --no-configs --no-exact-rows --generator-period 64 --steps 3 --warmup 1 > $O/bench_prof.log 2>&1
cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python3 $R/bench.py --rows 131072 --evals 2048 --no-cpu-baseline --no-configs --no-exact-rows --steps 1 --warmup 0 --generator-period 64 > $O/pmc_$C.log 2>&1
done
python3 $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) k_factor7 8192 2048 60 $O/r02_traffic.json
echo "== cfg4"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4_prof -- python3 $R/tools/configs.py cfg4 > $O/cfg4_prof.log 2>&1
cp $(find $O/cfg4_prof -name "*kernel_stats.csv" | head -1) $O/r02_cfg4_kernel_stats.csv
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/cfg4_pmc$i -- python3 $R/tools/batch_once.py 32768 40 512 > $O/cfg4_pmc$i.log 2>&1
  python3 $R/tools/pmc_rows.py $(find $O/cfg4_pmc$i -name "*counter_collection.csv" | head -1) k_factorw $((32768*512)) >> $O/r02_cfg4_sq_counters.txt
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/cfg4_pmc_$C -- python3 $R/tools/batch_once.py 32768 40 512 > $O/cfg4_pmc_$C.log 2>&1
done
python3 $R/tools/pmc_traffic.py $(find $O/cfg4_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/cfg4_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) k_factorw 8192 512 80 $O/r02_cfg4_traffic.json
echo "== cfg5"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5_prof -- python3 $R/tools/configs.py cfg5 > $O/cfg5_prof.log 2>&1
cp $(find $O/cfg5_prof -name "*kernel_stats.csv" | head -1) $O/r02_cfg5_kernel_stats.csv
echo "== solar"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solar_prof -- python3 $R/tools/solar_latency.py 100000 > $O/solar_prof.log 2>&1
cp $(find $O/solar_prof -name "*kernel_stats.csv" | head -1) $O/r02_solar_kernel_stats.csv
echo "== tools"
( echo "== tools/solar_latency.py 100000 1000000"; timeout -k 10 300 python3 $R/tools/solar_latency.py 100000 1000000
  echo "== tools/latency.py 1000000 30 1"; TP_CHUNKS=2048,1024,512 timeout -k 10 300 python3 $R/tools/latency.py 1000000 30 1
  echo "== tools/variance_latency.py"; timeout -k 10 300 python3 $R/tools/variance_latency.py
  echo "== tools/batch_once.py (wide kernel, us per row)"; for B in 256 512 768; do timeout -k 10 100 python3 $R/tools/batch_once.py 65536 40 $B 2; done; timeout -k 10 100 python3 $R/tools/batch_once.py 16384 86 256 2 ) > $O/r02_tools_output.txt 2>&1
rm -rf $O/*_prof $O/pmc_* $O/cfg4_pmc*
ls $O
