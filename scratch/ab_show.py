import json, sys
tag = sys.argv[1]
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print(tag, "FAILED", e); sys.exit(0)
k = d["kernels"]
g = lambda n: k.get(n, {}).get("ms")
print("%-14s step %.3f ms | tfwd %s tbwd %s | collect %s sort %s reg %s bsb %s red %s | R6 %s R7 %s | hits %s found %s" % (
    tag, d["ms_per_step"], g("trace_fwd"), g("trace_bwd"), g("trace.collect_hits"), g("trace.sort_composite_fwd"), g("trace.register_hits"),
    g("trace.batch_surfel_bwd"), g("trace.reduce_surfel_records"), g("composite_fwd"), g("composite_bwd"),
    (d.get("trace_counts") or {}).get("hits"), (d.get("trace_counts") or {}).get("found")))
tc = d.get("trace_counts") or {}
print("%-14s   nodes %s leaves %s entries %s max_list %s rows %s coop %s" % ("", tc.get("packet_nodes"), tc.get("packet_leaves"), tc.get("entries"), tc.get("max_list"), tc.get("compact_rows"), tc.get("coop_cycles")))
