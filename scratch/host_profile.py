"""Host-side cost of one EnvGS step (enqueue only): cProfile over 30 steps of bench.py's own step()."""
import cProfile, pstats, sys, os, io, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-render", "--steps", "30", "--warmup", "5"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
