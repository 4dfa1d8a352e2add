import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from cockroach_amd import _lib, device as dev
from workloads import tpch_plans as P, tpch_synth as T
_lib.check(_lib.load().crgpu_init(0))
t, rows = T.generate(10.0, "cuda:0", 42)
torch.cuda.synchronize()
class TimedExec(P.GpuExec):
    pass
def wrap(name):
    orig = getattr(P.GpuExec, name)
    def f(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        _lib.load().crgpu_dev_synchronize()
        self.times.append((name, (time.perf_counter() - t0) * 1e3))
        return r
    setattr(TimedExec, name, f)
for nme in ["select_i64", "select_bytes", "filter", "join", "project", "extract_year", "agg"]:
    wrap(nme)
for rep in range(3):
    ex = TimedExec(); ex.times = []
    t0 = time.perf_counter()
    tabs = {k: ex.table(v) for k, v in t.items()}
    t1 = time.perf_counter()
    r = P.q9(ex, tabs)
    t2 = time.perf_counter()
    ex.close()
    t3 = time.perf_counter()
    print("q9 tables %.2f plan %.2f close %.2f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), [(n, round(x, 2)) for n, x in ex.times])
for rep in range(2):
    ex = TimedExec(); ex.times = []
    tabs = {k: ex.table(v) for k, v in t.items()}
    t1 = time.perf_counter()
    r = P.q3(ex, tabs)
    t2 = time.perf_counter()
    ex.close()
    print("q3 plan %.2f" % ((t2-t1)*1e3), [(n, round(x, 2)) for n, x in ex.times])
