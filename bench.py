#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: path-steps/sec at 1M paths x 250 steps; PV/CVA rel-error vs CPU ref.

Workload: correlated Vasicek + CIR++ (rho = 0.5) payer IRS CVA, 12.5y quarterly swap,
51 exposure dates x 5 Euler sub-steps = 250 steps, 1,048,576 main-simulation paths PER GPU (weak scaling: paths are
sharded, one process per GPU; the only exchange of a pass is ONE all-gather of the 32-byte accumulator records).
A "step" = one pass of the hot path over the batch: Philox + Box-Muller + SDE sub-steps, cashflows / regression exposure,
CVA reduction (+ the collective).  The LSM pre-simulation (131,072 paths per GPU) runs once before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--paths P] [--scaling weak|strong]

`value` is the figure of --scaling (default weak: P paths PER GPU, the per-GPU work is fixed as N grows).  The metric is quoted on
1 M paths IN TOTAL, so the same line also carries `strong`: the same K steps timed on 1,048,576 paths split over the N GPUs
(131,072 per GPU at N = 8), and `sustained`: a second timed region of >= 1 s of passes (K steps of 1 ms are 20 ms of work).

--gpus N > 1 without a torch.distributed environment: this process starts N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`) BEFORE
touching the GPU and relays rank 0's JSON line; launched under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "montecarlo-risk-engine_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HAZARDS = {0.5: 0.006402303360855854, 1.0: 0.01553038972325307, 2.0: 0.009729741230773657, 3.0: 0.015552544648116201,
           4.0: 0.021196186202801115, 5.0: 0.02284319986706472, 7.0: 0.010111423894480876, 10.0: 0.00613267811172937,
           15.0: 0.0036969930706003337, 20.0: 0.003791311459217732}
HBM_PEAK_GBS = 8000.0                               # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
STRONG_TOTAL_PATHS = 1 << 20                       # the metric's configuration: 1 M paths x 250 steps in total
REF_CVA, REF_CVA_SE = 0.004623, 0.000012            # the reference itself on this workload at 50 k + 50 k paths
KERNEL_SOURCES = ["kf_lean.hip", "kf_common.h", "mcx_device.h", "mcx_math.h", "mcx_tables.h", "k1_paths.hip"]   # what the timed kernels are compiled from


def build_controller(n_main, n_pre, backend):
    import numpy as np
    from mcx.common.enums import SimulationScheme
    from mcx.controller.controller import SimulationController
    from mcx.metrics.cva_metric import CVAMetric
    from mcx.metrics.risk_metrics import RiskMetrics
    from mcx.models.cirpp import CIRPPModel
    from mcx.models.model_config import ModelConfig
    from mcx.models.vasicek import VasicekModel
    from mcx.products.netting_set import NettingSet
    from mcx.products.swap import InterestRateSwap, IRSType
    ir = VasicekModel(0.0, rate=0.03, mean=0.05, mean_reversion_speed=0.1, volatility=0.01, asset_id="irs")
    cr = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4)
    model = ModelConfig([ir, cr], inter_asset_correlation_matrix=np.array([0.5]))
    irs = InterestRateSwap(0.0, 12.5, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER, "irs")
    ns = [NettingSet(name="irs", products=[irs], counterparty_id="cp")]
    rm = RiskMetrics([CVAMetric("cp", 0.4)], exposure_timeline=np.arange(51) * 0.25)
    return SimulationController(ns, model, rm, n_main, n_pre, 5, SimulationScheme.EULER, backend=backend)


HOST_THREADS = [1]          # torch's intra-op thread count before main() takes it to one (restored for the CPU baseline)


def kernel_source_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "montecarlo-risk-engine_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(target_seconds=12.0):
    """the CPU oracle (oracle/mcx_oracle.c, OpenMP over paths) on a bounded sample of the same workload; returns the baseline
    object and (sample paths, pre-simulation paths, oracle CVA) for the GPU-vs-CPU comparison on identical Philox counters"""
    import torch
    from oracle_backend import OracleBackend
    be = OracleBackend()
    torch.set_num_threads(HOST_THREADS[0])        # the oracle's OpenMP loops share the runtime main() set to one thread: all cores here
    threads = int(be.lib.orc_num_threads())
    n_pre = 16384

    def run(n):
        sc = build_controller(n, n_pre, be)
        sc.prepare()
        t0 = time.perf_counter()
        res = sc.main_pass()
        return time.perf_counter() - t0, sc.sim_plan.n_steps, res[0][0][0]

    t, S, _ = run(16384)
    rate = 16384 * S / t
    n = int(min(1 << 20, max(16384, rate * target_seconds / S)))
    n = (n // 4096) * 4096
    t, S, cva = run(n)
    torch.set_num_threads(1)
    obj = {"value": n * S / t, "unit": "path-steps/s", "cores": threads, "kind": "port",
           "sample": f"{n} paths x {S} steps of the same workload (path generation + book + CVA on the CPU oracle, OpenMP {threads} threads), {t:.2f} s"}
    return obj, (n, n_pre, cva)


def launch_ranks(args, argv):
    """parent of an N-GPU run: never touches the GPU; starts the ranks, relays rank 0's JSON line, returns their exit code"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(proc.stdout)
    return proc.returncode if lines or proc.returncode else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths", type=int, default=1 << 20, help="main-simulation paths PER GPU (weak) / in total (strong)")
    ap.add_argument("--presim", type=int, default=131072, help="pre-simulation (LSM) paths PER GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the extra timed region on 1,048,576 paths in total")
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of the second (long) timed region; 0: off")
    ap.add_argument("--clock-warmup", type=float, default=1.0, help="seconds of untimed passes before the W warm-up steps (power state)")
    ap.add_argument("--plan", default="auto", choices=["auto", "semi", "fused", "unfused"],
                    help="main-pass execution plan: fused = one launch; semi = K1 + one book/metric kernel; unfused = K1,K2,K4")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node == --gpus\n")
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    # the host side of a pass is a few small torch / numpy operations: one intra-op thread (a pool sized from the core count of the
    # host spins a container with a CPU quota into CFS throttling: mcx/helpers/host_threads.py)
    HOST_THREADS[0] = torch.get_num_threads()
    torch.set_num_threads(1)
    torch.cuda.set_device(local_rank)
    grouped = "WORLD_SIZE" in os.environ                # under torch.distributed.run (also with one rank): RCCL process group
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)

    from mcx import _native
    be = _native.HipBackend(local_rank)
    per_gpu = args.paths if args.scaling == "weak" else args.paths // world
    pre_gpu = args.presim if args.scaling == "weak" else max(args.presim // world, 4096)
    sc = build_controller(per_gpu * world, pre_gpu * world, be)
    t0 = time.perf_counter()
    sc.prepare()
    be.synchronize()
    t_prepare = time.perf_counter() - t0
    S = sc.sim_plan.n_steps
    n_local = sc._main_engine.num_paths
    T, D, E = sc.sim_plan.n_dates, sc.sim_plan.n_state, len(sc.exposure_timeline)
    paths_buf = None

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # three execution plans exist for the main pass; time each once (N = 1) and keep the fastest; N > 1 runs the one-launch plan
    fused_obj = sc._fused
    names = ["fused", "semi", "unfused"] if args.plan == "auto" else [args.plan]
    if world > 1 and args.plan == "auto":
        names = ["fused"]
    if fused_obj is None:
        names = ["unfused"]
    if any(n != "fused" for n in names):
        paths_buf = be.empty_paths(T, D, n_local)

    def set_plan(name):
        sc._fused = None if name == "unfused" else fused_obj
        sc.main_plan = name if name != "unfused" else sc.main_plan

    plan_ms = {}
    for name in names:
        set_plan(name)
        sc.main_pass(paths_buf if name != "fused" else None)
        barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            sc.main_pass(paths_buf if name != "fused" else None)
        barrier()
        plan_ms[name] = (time.perf_counter() - t0) / 2 * 1e3
    best = min(plan_ms, key=plan_ms.get)
    set_plan(best)
    res = None
    # A GPU that sat idle through minutes of imports is in a low power state and needs about a second of load to reach its
    # clocks; W warm-up passes of ~1 ms each do not get it there.  Run the same pass untimed for --clock-warmup seconds first.
    # (a fixed pass count agreed by all ranks: every pass holds a collective, a time-based loop would not match across ranks)
    def agree(n):
        if grouped:
            tc = torch.tensor([n], dtype=torch.int64, device="cuda")
            dist.broadcast(tc, 0)
            n = int(tc.item())
        return n

    n_clock = agree(int(args.clock_warmup / max(plan_ms[best] * 1e-3, 1e-4)))
    for _ in range(n_clock):
        sc.main_pass(paths_buf if best != "fused" else None)
    fused = best == "fused"

    def timed_passes(sc, n_steps, warmup, want_kernel_ms):
        """W untimed + exactly n_steps timed passes of controller `sc` between barriers; returns (seconds [max over ranks], last
        result, kernel ms or nan).  One-launch plan: two passes in flight — the record gather over RCCL (under a process group), the
        copy to the host and the merge of pass k run while the kernel of pass k+1 computes (controller.fused_pass_begin /
        fused_pass_end); every pass still delivers its merged result inside the timed region.  One GPU without a group: the
        records land in pinned host memory directly; the same code path as N > 1."""
        res = None
        for _ in range(warmup):
            res = sc.main_pass(paths_buf if best != "fused" else None)
        # device time of the dominant kernel with HIP events on the launch stream (torch's current stream = the stream the
        # library launches on, _native.HipBackend._stream)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps if want_kernel_ms else 0)]
        pipelined = fused and sc.pipelined_passes_available() and not os.environ.get("MCX_BENCH_NO_PIPELINE")
        pending = None
        # the one-launch kernel is timed by the library itself: HIP event pairs recorded around its launch on the launch stream
        # (mcx_fused_set_timing), so that the figure is the kernel alone and not kernel + the 5 us record merge that follows it
        lib_timing = want_kernel_ms and fused and hasattr(be, "fused_set_timing")
        if lib_timing:
            # every 4th timed step carries the event pair (a pair costs ~8 us of stream time per step: 1.088 vs 1.079 ms measured)
            be.fused_set_timing(sc._fused, 4 if n_steps >= 8 else 1)
        barrier()
        t0 = time.perf_counter()
        for k in range(n_steps):
            if want_kernel_ms and not lib_timing:
                ev[k][0].record()
            if pipelined:
                ticket = sc.fused_pass_begin()
                if pending is not None:
                    res = sc.fused_pass_end(pending)
                pending = ticket
            elif fused:
                res = sc._fused_pass()               # one launch (+ block merge, record copy, rank gather)
                if want_kernel_ms and not lib_timing:
                    ev[k][1].record()
            elif best == "semi":
                paths = sc._main_engine.generate_paths_native(out=paths_buf)
                if want_kernel_ms:
                    ev[k][1].record()                # K1 device time; then ONE kernel for book + metrics
                res = sc._finish_fused_records(be.fused_eval_paths(sc._fused, paths))
            else:
                paths = sc._main_engine.generate_paths_native(out=paths_buf)
                if want_kernel_ms:
                    ev[k][1].record()
                cfs, expo = be.eval_book(sc.book, paths)
                res = sc._evaluate_all(sc._shard, cfs, expo, paths)
        if pending is not None:
            res = sc.fused_pass_end(pending)
        barrier()
        dt = time.perf_counter() - t0
        if grouped:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        k_ms = float("nan")
        if lib_timing:
            tk = be.fused_kernel_times(sc._fused)
            k_ms = float(np.mean(tk)) if tk.size else float("nan")        # (the first 64 steps when --steps is larger)
            be.fused_set_timing(sc._fused, 0)
        elif want_kernel_ms:
            k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return dt, res, k_ms, pipelined

    dt, res, k1_ms, pipelined = timed_passes(sc, args.steps, args.warmup, True)
    # a second timed region of >= 1 s of the same passes: K = 20 steps of 1 ms are 20 ms, too short for an outside clock to check
    n_long = agree(max(args.steps, int(math.ceil(args.sustain / max(dt / args.steps, 1e-5))))) if args.sustain > 0 else 0
    sustained = None
    if n_long:
        dt_long, _, _, _ = timed_passes(sc, n_long, 0, False)
        sustained = {"steps": n_long, "seconds": dt_long, "ms_per_step": dt_long / n_long * 1e3,
                     "value": (sc._main_engine.num_paths * world if world == 1 else per_gpu * world) * S * n_long / dt_long}
    # the metric's own configuration: 1,048,576 paths IN TOTAL over the N GPUs (strong scaling), same K steps, same loop
    strong = None
    if args.scaling == "weak" and fused and not args.no_strong and (world > 1 or per_gpu != STRONG_TOTAL_PATHS):
        sp, spre = STRONG_TOTAL_PATHS // world, max(args.presim // world, 4096)
        sc_s = build_controller(sp * world, spre * world, be)
        sc_s.prepare()
        sc_s.main_pass()
        dt_s, res_s, k_s, _ = timed_passes(sc_s, args.steps, args.warmup, True)
        strong = {"total_paths": sp * world, "paths_per_gpu": sp, "ms_per_step": dt_s / args.steps * 1e3, "kernel_ms": k_s,
                  "value": sp * world * S * args.steps / dt_s, "cva": res_s[0][0][0][0], "mc_error": res_s[0][0][0][1]}
        del sc_s

    if rank == 0:
        cva, err = res[0][0][0]
        total_paths = n_local * world if world == 1 else per_gpu * world
        value = total_paths * S * args.steps / dt
        pass_bytes = 8.0 * (2 * T * D + 2 * E + 2) * n_local   # SURVEY.md §8d B_path for the whole pass
        kname = ("kf_lean<2,2,SIG_VAS_CIR_E,PPL=2> (Philox + Box-Muller + SDE + LSM exposure + CVA in one launch)" if fused else
                 "k1_paths<2,2,SIG_VAS_CIR_E> (Philox4x32-10 + Box-Muller + Cholesky + Vasicek/CIR++ Euler)")
        k1_bytes = pass_bytes if fused else 8.0 * T * D * n_local
        achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
        # counters of the dominant kernel (tools/measure_traffic.sh, tools/measure_sq.sh: separate rocprofv3 --pmc passes at
        # 2^20 paths) — used only when they were taken from the kernel sources this library was built from
        sha = kernel_source_sha()
        traffic, alu, counters_note = None, None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
            if pm.get("kernel_source_sha") == sha:
                c = pm["kf_lean" if fused else "k1_paths"]
                traffic = c["hbm_bytes_per_launch_at_1Mi_paths"] * (n_local / float(1 << 20))
                # VALU-issue model: wave-instructions per launch by class (SQ_INSTS_VALU total; the class mix of the sub-step loop
                # from its ISA, tools/asm_mix.py; everything outside the sub-step loop is f64 polynomial work) x the SUSTAINED
                # issue cost of the class measured by tools/ubench_valu on this chip, / 1024 SIMDs
                per_ps = c["valu_insts_per_wave"] * c["waves"] * 64.0 / (float(1 << 20) * S)           # VALU per path and sub-step
                mix = dict(pm["substep_mix_per_path"]) if fused else None
                if mix:
                    cls = {k: mix[k] for k in ("mad_u64_u32", "int32", "rsq_f64", "minmax_f64")}
                    cls["f64"] = per_ps - sum(cls.values())
                    ns = pm["issue_ns"]
                    model_ms = sum(cls[k] * ns[k] for k in cls) * (n_local / 64.0) * S / 1024.0 * 1e-6
                    alu = {"bound": "f64 VALU issue", "valu_insts_per_path_step": per_ps, "mix_per_path_step": cls,
                           "issue_ns_per_wave_instruction_per_simd": ns, "modelled_ms": model_ms, "frac": model_ms / k1_ms,
                           "note": "issue costs are the SUSTAINED rates tools/ubench_valu measures with every CU busy: HIP events around ONE "
                                   ">= 10 ms launch per instruction class, 8 waves per SIMD (profiles/r03_ubench_valu.txt; the shader clock "
                                   "during those launches, 2.15-2.39 GHz by class, is in profiles/r03_ubench_valu_clock_mhz.txt: v_fma_f64 "
                                   "1.94 ns = 4.2 clk at the 2.18 GHz it holds); frac = modelled VALU issue time / kernel time — what is left "
                                   "is scalar work and waits the four waves of a SIMD do not fully overlap",
                           "source": "profiles/counters.json (tools/measure_counters.sh)"}
            else:
                counters_note = "profiles/counters.json was measured on other kernel sources: traffic / alu omitted"
        except Exception:
            counters_note = "profiles/counters.json missing: traffic / alu omitted"
        out = {
            "metric": "path-steps/sec at 1M paths x 250 steps; PV/CVA rel-error vs CPU ref",
            "value": value, "unit": "path-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Vasicek+CIR++ (rho=0.5) payer IRS CVA, Euler, 51 dates x 5 sub-steps (SURVEY §8d config 3)",
                       "paths_per_gpu": n_local, "steps_per_path": S, "state_dim": D, "stored_dates": T,
                       "exposure_dates": E, "presim_paths_per_gpu": pre_gpu, "parallelism": f"paths x{world}",
                       "total_paths": total_paths, "value_is": f"{args.scaling} scaling: {total_paths} paths in total ({n_local} per GPU)",
                       "execution_plan": best, "plan_probe_ms": plan_ms,
                       "passes_in_flight": 2 if pipelined else 1},
            "strong": strong, "sustained": sustained,
            "roofline": {"bound": "hbm", "binding_resource": "f64 VALU issue (see `alu`)" if fused else "f64 VALU issue", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": k1_ms,
                         "algorithmic_bytes_per_launch": k1_bytes,
                         "note": "achieved = ALGORITHMIC bytes of the pass (SURVEY §8d: 4096 B/path, what the unfused dataflow "
                                 "moves) / kernel time, i.e. an HBM-roofline EQUIVALENT: the fused kernel materialises none of these bytes "
                                 "(`traffic` = measured HBM bytes) and the resource that binds it is f64 VALU issue — `alu.frac` is the "
                                 "fraction of THAT bound"},
            "alu": alu,
            "result": {"cva": cva, "mc_error": err, "z_vs_reference": (cva - REF_CVA) / math.hypot(err, REF_CVA_SE),
                       "reference_cva": REF_CVA, "reference_mc_error": REF_CVA_SE},
            "prepare_s": t_prepare,
        }
        if counters_note:
            out["counters_note"] = counters_note
        if not args.no_cpu_baseline and world == 1:
            cb, (n_s, n_pre_s, cva_cpu) = cpu_baseline()
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]
            # the reference's own PyTorch-CPU path cannot be timed here (/root/reference does not exist on the GPU box): the survey's
            # figure for this configuration at 1/10 of the paths, measured in the build container, for orientation only
            out["reference_cpu_survey"] = {"value": 1.27e7, "unit": "path-steps/s (main path generation alone; 4.0e6 end to end)", "cores": 8,
                                           "where": "BASELINE.md section 2: 100k + 100k paths, 8 vCPU Xeon 2.1 GHz, build container"}
            # PV/CVA vs the CPU reference path: the same sample on the GPU, identical Philox counters
            sc_s = build_controller(n_s, n_pre_s, be)
            sc_s.prepare()
            cva_gpu_s, err_gpu_s = sc_s.main_pass()[0][0][0]
            out["result"].update({"sample_paths": n_s, "cva_gpu_at_sample": cva_gpu_s, "cva_cpu_at_sample": cva_cpu[0],
                                  "rel_error_vs_cpu": abs(cva_gpu_s - cva_cpu[0]) / abs(cva_cpu[0]),
                                  "mc_error_rel_diff_vs_cpu": abs(err_gpu_s - cva_cpu[1]) / abs(cva_cpu[1])})
        print(json.dumps(out), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
