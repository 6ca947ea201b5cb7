import json, sys
for n in sys.argv[1:]:
    for l in open('gpurun_out/r02_coopv_%s.json' % n):
        if l.startswith('{'):
            d = json.loads(l); k = d['kernels']; tc = d['trace_counts']
            print(n, d['ms_per_step'], {a.split('.')[-1][:10]: k[a]['ms'] for a in ('trace_fwd', 'trace.collect_hits', 'trace.sort_composite_fwd') if a in k}, tc['found'], tc['packet_nodes'], tc['max_list'], tc.get('coop_cycles'))
