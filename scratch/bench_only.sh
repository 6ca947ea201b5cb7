# the bench lines of profiles/collect_profiles.sh without the profiler passes (gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python bench.py > $O/bench_envgs_final.json 2> $O/bench_envgs_final.err
python bench.py --caller reference --no-cpu-baseline > $O/bench_envgs_reference_caller_final.json 2> $O/bench_envgs_reference_caller_final.err
python bench.py --caller twin --no-cpu-baseline > $O/bench_envgs_twin_caller_final.json 2> $O/bench_envgs_twin_caller_final.err
python bench.py --workload raster > $O/bench_raster_final.json 2> $O/bench_raster_final.err
python bench.py --env-gaussians 700000 --no-cpu-baseline --steps 15 --warmup 4 > $O/bench_env700k_final.json 2> $O/bench_env700k_final.err
python bench.py --feature-dtype f16 --no-cpu-baseline > $O/bench_envgs_f16_final.json 2> $O/bench_envgs_f16_final.err
python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --no-cpu-baseline --steps 8 --warmup 3 > $O/bench_config5_final.json 2> $O/bench_config5_final.err
