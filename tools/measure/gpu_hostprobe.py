"""Where does host time go per step?  (graph vs eager; per-call enqueue cost)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import Workload
from siammask_amd import _lib

dev = torch.device("cuda", 0)
for graph in (True, False):
    w = Workload("sharp_b8_f16", dev, 0)
    w.model._graph = graph
    _lib.check(_lib.lib().smk_set_graph_mode(w.model._ctx, 1 if graph else 0))
    for i in range(10):
        w.step(i)
    torch.cuda.synchronize()
    N = 100
    t0 = time.perf_counter()
    for i in range(N):
        w.step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("graph=%d  host enqueue %.1f us/step   total %.1f us/step" % (graph, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    m = w.model
    x = w.xs[0]
    def timeit(fn, n=200):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        e = time.perf_counter(); torch.cuda.synchronize()
        return (e - t) / n * 1e6
    print("   track_mask enqueue %.1f us, track_refine enqueue %.1f us" % (timeit(lambda: m.track_mask(x)), timeit(lambda: m.track_refine(w.pos))))
    L = _lib.lib()
    sp = _lib.current_stream_ptr()
    cls = m._io.get("cls") if graph else torch.empty(8, 10, 25, 25, device=dev)
    loc = m._io.get("loc") if graph else torch.empty(8, 20, 25, 25, device=dev)
    mk = m._io.get("mask") if graph else torch.empty(8, 3969, 25, 25, device=dev)
    xin = m._io.get("x") if graph else x
    print("   raw smk_track call %.1f us" % timeit(lambda: L.smk_track(m._ctx, xin.data_ptr(), 8, 1, cls.data_ptr(), loc.data_ptr(), mk.data_ptr(), sp)))
    print("   current_stream_ptr %.1f us, cuda.device ctx %.1f us, copy_ %.1f us, clone %.1f us" % (
        timeit(lambda: _lib.current_stream_ptr()), timeit(lambda: torch.cuda.device(0).__enter__()),
        timeit(lambda: xin.copy_(x)) if graph else 0.0, timeit(lambda: cls.clone())))
    del w
