# A/B runs of the cooperative collection (gpurun): python bench lines + trace counts
cd $GRAFT_REPO_ROOT
for v in $*; do
  python bench.py --no-cpu-baseline --no-render --steps 20 --warmup 5 --debug-trace $v > gpurun_out/r02_coopv_$v.json 2>/dev/null
done
