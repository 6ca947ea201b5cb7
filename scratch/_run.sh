cd $GRAFT_REPO_ROOT
for dt in 0 32768 65536; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --env-gaussians 700000 --steps 15 --warmup 4 --debug-trace $dt 2>/dev/null | python scratch/ab_show.py "e700k.rg$dt" | grep -v "nodes "
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --gaussians 1800000 --env-gaussians 630000 --steps 10 --warmup 3 --debug-trace $dt 2>/dev/null | python scratch/ab_show.py "caps.rg$dt" | grep -v "nodes "
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --steps 8 --warmup 3 --debug-trace $dt 2>/dev/null | python scratch/ab_show.py "c5.rg$dt" | grep -v "nodes "
done
