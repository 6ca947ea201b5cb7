# Run on the GPU box (gpurun):  bash profiles/collect_profiles.sh [keys...]   -> raw outputs under gpurun_out/, then `python profiles/summarize.py r06` here.
# keys (default: all): envgs raster config5 env700k base_trace_d0 base_trace_d2 -- one tracked counter summary per WORKLOAD (profiles/r06_pmc_<key>.json);
# bench.py quotes the counters of the workload it is running and nothing else (bench.py:workload_key).
# Every command is bounded by `timeout` (round 3 lost 15 GPU-minutes to a counter pass that aborted and then hung).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
KEYS="${@:-envgs raster config5 env700k base_trace_d0 base_trace_d2}"
COMMON="--no-cpu-baseline --no-render --no-reference-caller"
args_of() {
  case $1 in
    envgs) echo "" ;;
    raster) echo "--workload raster" ;;
    config5) echo "--height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16" ;;
    env700k) echo "--env-gaussians 700000" ;;
    base_trace_d0) echo "--workload base_trace --trace-depth 0" ;;
    base_trace_d2) echo "--workload base_trace --trace-depth 2" ;;
  esac
}
for K in $KEYS; do
  A="$(args_of $K)"
  B="python $R/bench.py $COMMON $A"
  rm -rf $O/p_$K $O/pmc_${K}_*
  ST=20; [ $K = config5 ] && ST=6; [ $K = raster ] && ST=30
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$K -o $K -- $B --steps $ST --warmup 4 --repeats 3 > /dev/null 2>&1
  # counters: one pass each (TCC: FETCH_SIZE / WRITE_SIZE cannot share a pass; SQ: <= 8 per pass); --pmc only with --kernel-trace
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${K}_fetch -o f -- $B --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${K}_write -o w -- $B --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_${K}_insts -o c -- $B --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_${K}_active -o c -- $B --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d $O/pmc_${K}_tcp -o c -- $B --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
  # keep only what summarize.py reads (the raw traces are tens of MB; gpurun merges <= 64 MiB back)
  find $O/p_$K -type f ! -name "*kernel_stats.csv" -delete
  for d in $O/pmc_${K}_*; do find $d -type f ! -name "*counter_collection.csv" -delete; done
  python $R/profiles/shrink_counters.py $O/pmc_${K}_*
  echo "collected $K: $(ls $O/pmc_${K}_* 2>/dev/null | wc -l) files"
done
case " $KEYS " in *" envgs "*)
  # kernel inventory of one default step (bench.py's `launches` field reads the summary of it)
  rm -rf $O/p_inv
  timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/p_inv -o inv -- python $R/bench.py $COMMON --steps 6 --warmup 3 --repeats 1 > /dev/null 2>&1
  cd $R
  python scratch/step_inventory.py $(find $O/p_inv -name "*kernel_trace.csv" | head -1) 30 > $O/step_inventory.txt 2>&1; python scratch/trace_gaps.py $(find $O/p_inv -name "*kernel_trace.csv" | head -1) >> $O/step_inventory.txt 2>&1
  rm -rf $O/p_inv ;;
esac
