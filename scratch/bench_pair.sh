#!/bin/bash
# quick look at a build:  bash scratch/bench_pair.sh   (EnvGS step x2, raster-only, both caps) -- per-kernel ms of the binning kernels included
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B="python bench.py --no-cpu-baseline --no-render --no-reference-caller"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]; g=lambda n:(k.get(n) or {}).get("ms")
print(sys.argv[1], "step %.3f ms"%d["ms_per_step"], {n:g(n) for n in ("project_surfels","scan_tiles_touched","bin_tile_pairs","sort_tile_lists","composite_fwd","composite_bwd","trace_fwd","trace_bwd")}, "N", (d.get("raster_stats") or d.get("config",{})).get("N"))'
for rep in 1 2; do timeout 100 $B --steps 20 --warmup 5 2>/dev/null | python -c "$show" envgs.$rep; done
timeout 100 $B --workload raster --steps 30 --warmup 5 2>/dev/null | python -c "$show" raster
timeout 150 $B --gaussians 1800000 --env-gaussians 630000 --steps 10 --warmup 3 2>/dev/null | python -c "$show" caps
