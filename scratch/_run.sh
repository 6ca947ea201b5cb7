cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-render --no-reference-caller"
for rep in 1 2; do
timeout 300 python bench.py --workload base_trace --trace-depth 2 $Q --steps 10 --warmup 3 2>/dev/null | python scratch/ab_show.py "bt2.defer.$rep" | grep -v "nodes "
timeout 300 python bench.py --workload base_trace --trace-depth 2 $Q --steps 10 --warmup 3 --no-deferred-surfel-grads 2>/dev/null | python scratch/ab_show.py "bt2.plain.$rep" | grep -v "nodes "
timeout 300 python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 $Q --steps 8 --warmup 3 2>/dev/null | python scratch/ab_show.py "c5.$rep" | grep -v "nodes "
python bench.py $Q --steps 20 --warmup 5 2>/dev/null | python scratch/ab_show.py "envgs.$rep" | grep -v "nodes "
done
