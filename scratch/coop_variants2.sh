cd $GRAFT_REPO_ROOT
for sg in 2 3 4; do for v in 0 4096; do
  python bench.py --no-cpu-baseline --no-render --steps 20 --warmup 5 --debug-trace $v --debug-segments $sg > gpurun_out/r02_coopv_s${sg}_$v.json 2>/dev/null
done; done
