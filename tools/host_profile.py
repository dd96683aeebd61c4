"""Where the host time of one solve!(model) goes (Python host, do-nothing optimizer): cProfile over repeated solves of a mid-size model.
python tools/host_profile.py [n r m]"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mid_table as M
import parametron_jl_amd as P
n, r, m = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 150, 30)
model, bufs, rng = M.build(n, r, m, host=True)
pre = [{k: rng.random(a.shape) for k, a in bufs.items()} for _ in range(2)]
def once(it):
    for k, a in bufs.items():
        a[...] = pre[it & 1][k]
    P.solve(model)
for it in range(20):
    once(it)
pr = cProfile.Profile()
pr.enable()
for it in range(300):
    once(it)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print("\n".join(l[:170] for l in out.getvalue().splitlines()[:50]))
