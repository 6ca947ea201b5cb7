cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_envgs $O/p_raster $O/pmc_fetch $O/pmc_write $O/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_envgs -o envgs -- python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-render > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_raster -o raster -- python $R/bench.py --workload raster --steps 30 --warmup 4 --no-cpu-baseline --no-render > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render > /dev/null 2>&1
cd $R
python bench.py > $O/bench_envgs_final.json 2> $O/bench_envgs_final.err
python bench.py --workload raster > $O/bench_raster_final.json 2> $O/bench_raster_final.err
# SURVEY.md 8(d)'s other sizes: the env set at its 700 000-surfel cap, and a configs[4]-like run (1200x1600, -ch07 raster, two specular bounces)
python bench.py --env-gaussians 700000 --no-cpu-baseline --steps 15 --warmup 4 > $O/bench_env700k_final.json 2> $O/bench_env700k_final.err
python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --no-cpu-baseline --steps 8 --warmup 3 > $O/bench_config5_final.json 2> $O/bench_config5_final.err
ls $O/p_envgs $O/pmc_fetch | head
