# end-of-round profile set of round 4 (run from the repo root on the GPU box through gpurun):
#   bash tools/profiles_r04.sh [bench|cfg2|cfg3|cfg3s|cfg4|cfg4s|cfg5|solar|tools ...]      (no argument: everything)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
WHAT="$*"; [ -z "$WHAT" ] && WHAT="bench cfg2 cfg3 cfg3s cfg4 cfg4s cfg5 solar tools"
prof() {  # name script args... -> $O/r04_NAME_kernel_stats.csv
  local name=$1; shift; local script=$R/$1; shift
  rm -rf $O/prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $script "$@" > $O/prof_$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/r04_${name}_kernel_stats.csv
  rm -rf $O/prof_$name
}
pmc() {   # tag counters script args... -> $O/pmc_TAG/...counter_collection.csv (path echoed)
  local tag=$1 ctr="$2"; shift 2; local script=$R/$1; shift
  rm -rf $O/pmc_$tag
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 $script "$@" > $O/pmc_$tag.log 2>&1
  find $O/pmc_$tag -name "*counter_collection.csv" | head -1
}
sq() {    # out kernel work script args... : the SQ counter groups, reduced per unit of work
  local out=$1 kern=$2 work=$3; shift 3
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    f=$(pmc sqtmp "$C" "$@")
    python3 $R/tools/pmc_rows.py $f $kern $work >> $out
  done
  rm -rf $O/pmc_sqtmp $O/pmc_sqtmp.log
}
for w in $WHAT; do case $w in
bench)
  echo "== bench"; timeout -k 10 900 python3 $R/bench.py > $O/r04_bench.json 2> $O/bench.err; echo rc=$?
  prof bench bench.py --no-cpu-baseline --no-configs --no-exact-rows --generator-period 64 --steps 3 --warmup 1
  BA="bench.py --rows 131072 --evals 2048 --no-cpu-baseline --no-configs --no-exact-rows --steps 1 --warmup 0 --generator-period 64"
  f=$(pmc fetch FETCH_SIZE $BA); g=$(pmc write WRITE_SIZE $BA)
  python3 $R/tools/pmc_traffic.py $f $g k_factor7 8192 2048 60 $O/r04_traffic.json
  rm -f $O/r04_sq_k_factor7.txt
  sq $O/r04_sq_k_factor7.txt k_factor7 $((8192*2048)) $BA
  ;;
cfg2)
  echo "== cfg2: the drop-in class and one chain through the batched evaluator"
  prof cfg2_compute tools/cfg2_once.py compute
  prof cfg2_predict tools/cfg2_once.py predict
  prof b1_evaluator tools/b1_loop.py
  ;;
cfg3)
  echo "== cfg3"; prof cfg3 tools/configs.py cfg3
  rm -f $O/r04_sq_cfg3_factor7.txt $O/r04_sq_cfg3_phi7.txt
  # two sweeps at 8 chunks of 8128 rows (what bench.py's cfg3 leg runs): the nominal pass sweeps every row,
  # the transition sweep the chunks 1 ... 7
  sq $O/r04_sq_cfg3_factor7.txt "k_factor7<40" $((65000*256)) tools/cfg3_chunks.py 8128
  sq $O/r04_sq_cfg3_phi7.txt "k_phi7<40" $(((65000-8128)*256)) tools/cfg3_chunks.py 8128
  f=$(pmc fetch FETCH_SIZE tools/cfg3_chunks.py 8128); g=$(pmc write WRITE_SIZE tools/cfg3_chunks.py 8128)
  python3 $R/tools/pmc_bytes.py $f $g $O/r04_cfg3_traffic.json "k_factor7<40, true" "k_factor7<40, false" "k_phi7<40" k_combine k_corr_small
  ;;
cfg3s)
  echo "== cfg3 shard"; prof cfg3s tools/configs.py cfg3s
  ;;
cfg4)
  echo "== cfg4"; prof cfg4 tools/configs.py cfg4
  ;;
cfg4s)
  echo "== cfg4 shard"; prof cfg4s tools/configs.py cfg4s
  ;;
cfg5)
  echo "== cfg5"; prof cfg5 tools/configs.py cfg5
  f=$(pmc fetch FETCH_SIZE tools/configs.py cfg5); g=$(pmc write WRITE_SIZE tools/configs.py cfg5)
  python3 $R/tools/pmc_bytes.py $f $g $O/r04_cfg5_traffic.json "k_mmR_mfma<4, true" "k_mmR_mfma<4, false" k_lincombine_mm
  ;;
solar)
  echo "== solar"; prof solar tools/solar_latency.py 100000
  prof solar_compute tools/solar_compute_once.py 100000 3
  ;;
tools)
  echo "== tools"
  ( echo "== tools/solar_latency.py 100000 1000000"; timeout -k 10 300 python3 $R/tools/solar_latency.py 100000 1000000
    echo "== tools/solar_compute_once.py 100000 / 1000000"; timeout -k 10 200 python3 $R/tools/solar_compute_once.py 100000 3; timeout -k 10 200 python3 $R/tools/solar_compute_once.py 1000000 3
    echo "== tools/dense_latency.py 172"; timeout -k 10 200 python3 $R/tools/dense_latency.py 172
    echo "== tools/dense_latency.py 80"; timeout -k 10 200 python3 $R/tools/dense_latency.py 80
    echo "== tools/latency.py 1000000 30 1"; TP_CHUNKS=2048,1024,512 timeout -k 10 300 python3 $R/tools/latency.py 1000000 30 1
    echo "== tools/variance_latency.py"; timeout -k 10 300 python3 $R/tools/variance_latency.py
    echo "== tools/cfg3_chunks.py 8128 7232"; timeout -k 10 200 python3 $R/tools/cfg3_chunks.py 8128 7232
    echo "== cfg4 shard, chunk lengths (tools/batch_once.py 200000 40 64)"; for L in 25024 16704 8384; do CHUNK_LEN=$L timeout -k 10 200 python3 $R/tools/batch_once.py 200000 40 64 3; done
    echo "== tools/batch_once.py (wide kernel streamed, us per row)"; for B in 256 512; do timeout -k 10 100 python3 $R/tools/batch_once.py 65536 40 $B 2; done ) > $O/r04_tools_output.txt 2>&1
  ;;
esac; done
rm -rf $O/pmc_*
ls $O
