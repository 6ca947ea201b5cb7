# Run on the GPU box (gpurun):  bash profiles/collect_profiles.sh   -> raw outputs under gpurun_out/, then `python profiles/summarize.py r05` here.
# Every command is bounded by `timeout` (round 3 lost 15 GPU-minutes to a counter pass that aborted and then hung).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_envgs $O/p_raster $O/pmc_* $O/p_inv
B="python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_envgs -o envgs -- $B --steps 20 --warmup 4 > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_raster -o raster -- $B --workload raster --steps 30 --warmup 4 > /dev/null 2>&1
# counters: one pass each (TCC: FETCH_SIZE / WRITE_SIZE cannot share a pass; SQ: <= 8 per pass); --pmc only with --kernel-trace
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_insts -o c -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_active -o c -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d $O/pmc_tcp -o c -- $B --steps 3 --warmup 1 > /dev/null 2>&1
# kernel inventory of one step (bench.py's `launches` field reads the summary of it)
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/p_inv -o inv -- python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 6 --warmup 3 --repeats 1 > /dev/null 2>&1
cd $R
python scratch/step_inventory.py $(find $O/p_inv -name "*kernel_trace.csv" | head -1) 30 > $O/step_inventory.txt 2>&1; python scratch/trace_gaps.py $(find $O/p_inv -name "*kernel_trace.csv" | head -1) >> $O/step_inventory.txt 2>&1
python profiles/summarize.py r05 > /dev/null 2>&1     # the bench lines below quote the counters of THIS collection (profiles/r05_pmc_envgs.json)
timeout 400 python bench.py > $O/bench_envgs_final.json 2> $O/bench_envgs_final.err
timeout 120 python bench.py --caller reference --no-cpu-baseline > $O/bench_envgs_reference_caller_final.json 2> $O/bench_envgs_reference_caller_final.err
timeout 120 python bench.py --caller twin --no-cpu-baseline --no-reference-caller > $O/bench_envgs_twin_caller_final.json 2> $O/bench_envgs_twin_caller_final.err
timeout 200 python bench.py --workload raster > $O/bench_raster_final.json 2> $O/bench_raster_final.err
# SURVEY.md 8(d)'s other sizes: the env set at its 700 000-surfel cap, both sets near their caps, and BASELINE configs[4] (1200x1600, -ch07 raster, two specular bounces, fp16 storage)
timeout 120 python bench.py --env-gaussians 700000 --no-cpu-baseline --no-reference-caller --steps 15 --warmup 4 > $O/bench_env700k_final.json 2> $O/bench_env700k_final.err
timeout 150 python bench.py --gaussians 1800000 --env-gaussians 630000 --no-cpu-baseline --no-reference-caller --steps 10 --warmup 3 > $O/bench_caps_final.json 2> $O/bench_caps_final.err
timeout 120 python bench.py --feature-dtype f16 --no-cpu-baseline --no-reference-caller > $O/bench_envgs_f16_final.json 2> $O/bench_envgs_f16_final.err
bash scratch/bvh_prof.sh > $O/bvh_prof.txt 2>&1
timeout 200 python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --no-cpu-baseline --no-reference-caller --steps 8 --warmup 3 --step-times 6 > $O/bench_config5_final.json 2> $O/bench_config5_final.err
python -m pytest tests -q -m perf -s > $O/perf_final.log 2>&1
ls $O/p_envgs $O/pmc_fetch $O/pmc_insts | head
