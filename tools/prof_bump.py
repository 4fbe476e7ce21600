#!/usr/bin/env python3
"""wall time of the bump-and-revalue path of config 3 (17 controllers: compile, uploads, LSM, main pass each), per phase, and the
process CPU time and the CFS throttling of the container (cpu.stat) around every run.  What it found (round 3): the 60-80 ms
outliers in arbitrary phases — Python code as often as HIP calls — were the container being THROTTLED: torch's 128-thread intra-op
pool spinning through a 16-CPU quota (13.5 s of CPU time in a 0.8 s run).  Not the cyclic collector (0.3 ms per run, gc.callbacks),
not the runtime's interrupt waits (HSA_ENABLE_INTERRUPT=0 changes nothing).  The controller now plans on one intra-op thread
(mcx/helpers/host_threads.py); `unguarded` shows the old behaviour.

    python tools/prof_bump.py [default|unguarded|torch1|freeze|disable|profile]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "montecarlo-risk-engine_amd"))
import bench, torch
from mcx import _native
from mcx.controller.controller import SimulationController as SC
be = _native.HipBackend(0)
sc = bench.build_controller(1 << 20, 131072, be)
sc.differentiate = True
sc.forward_mode = False
log = []
def timed(name, f):
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); log.append((name, (time.perf_counter() - t0) * 1e3)); return r
    return g
for name in ("_compile_all", "_perform_regression", "main_pass"):
    setattr(SC, name, timed(name, getattr(SC, name)))
for name in ("sim_create", "book_create", "fused_create", "lsm_stats", "lsm_run", "generate_paths"):
    setattr(type(be), name, timed(name, getattr(type(be), name)))
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "torch1":
    torch.set_num_threads(1)
if mode == "unguarded":                       # the behaviour before the guard: the pool at its default size throughout
    import mcx.helpers.host_threads as ht
    ht.single_threaded_host.__enter__ = lambda self: setattr(self, "_n", 1) or self
gc_log, gc_t0 = [], [0.0]
def on_gc(phase, info):
    if phase == "start":
        gc_t0[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], (time.perf_counter() - gc_t0[0]) * 1e3))
gc.callbacks.append(on_gc)
res = sc.run_simulation()                     # warm-up (first uploads, allocator growth)
if mode == "freeze":
    gc.collect(); gc.freeze()
elif mode == "disable":
    gc.collect(); gc.disable()
print("collector:", mode, " tracked objects:", len(gc.get_objects()), flush=True)
def cpu_stat():
    """(throttled periods, throttled microseconds) of this process's CPU cgroup, or None"""
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            d = dict(l.split() for l in open(f).read().strip().splitlines())
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
        except Exception:
            continue
    return None
import threading
print("threads:", threading.active_count(), " torch threads:", torch.get_num_threads(), " cpus:", os.cpu_count(),
      " affinity:", len(os.sched_getaffinity(0)), flush=True)
for rep in range(3):
    log.clear(); gc_log.clear()
    cs0, pt0 = cpu_stat(), time.process_time()
    torch.cuda.synchronize(); t0 = time.perf_counter(); res = sc.run_simulation(); torch.cuda.synchronize()
    print("bumps %.1f ms   gc: %d collections, %.1f ms in total, oldest generation: %s ms" % (
        (time.perf_counter() - t0) * 1e3, len(gc_log), sum(ms for _, ms in gc_log), [round(ms, 1) for g, ms in gc_log if g == 2]), flush=True)
    cs1 = cpu_stat()
    print("   process CPU time %.1f ms;  cgroup throttling during the run: %s" % (
        (time.process_time() - pt0) * 1e3, None if cs0 is None else "%d periods, %.1f ms" % (cs1[0] - cs0[0], (cs1[1] - cs0[1]) / 1e3)), flush=True)
    agg = {}
    for n, ms in log:
        agg.setdefault(n, []).append(round(ms, 1))
    for n, v in agg.items():
        print("   %-20s sum %7.1f  %s" % (n, sum(v), v), flush=True)

if mode == "profile":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); sc.run_simulation(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
