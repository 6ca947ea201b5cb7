cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 2>/dev/null | python scratch/ab_show.py "lowprio.$rep" | grep -v "nodes "
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --debug-trace 32768 2>/dev/null | python scratch/ab_show.py "aux1.$rep" | grep -v "nodes "
done
