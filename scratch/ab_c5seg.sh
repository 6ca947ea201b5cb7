#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for sg in 2 3 4; do
  timeout 200 python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --no-cpu-baseline --no-reference-caller --no-render --steps 8 --warmup 3 --debug-segments $sg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('segments $sg', d['ms_per_step'], d['value'])"
done
