cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-render --no-reference-caller"
for rep in 1 2; do
for d in 0 2; do
timeout 300 python bench.py --workload base_trace --trace-depth $d $Q --steps 10 --warmup 3 2>/dev/null | python scratch/ab_show.py "bt$d.defer.$rep" | grep -v "nodes "
timeout 300 python bench.py --workload base_trace --trace-depth $d $Q --steps 10 --warmup 3 --no-deferred-surfel-grads 2>/dev/null | python scratch/ab_show.py "bt$d.plain.$rep" | grep -v "nodes "
done; done
