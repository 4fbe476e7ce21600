#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: path-steps/sec at 1M paths x 250 steps (per GPU).

Workload: correlated Vasicek + CIR++ (rho = 0.5) payer IRS CVA, 12.5y quarterly swap,
51 exposure dates x 5 Euler sub-steps = 250 steps, 1,048,576 main-simulation paths PER GPU (weak scaling: paths are
sharded, one process per GPU, the only exchange is the gather of the (n, shift, s1, s2) accumulator record).
A "step" = one pass of the hot path over the batch: K1 path generation -> K2 book evaluation -> K4 CVA reduction
(+ collective). The LSM pre-simulation (131,072 paths) runs once before the timed region and is reported separately.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--paths P]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "montecarlo-risk-engine_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

HAZARDS = {0.5: 0.006402303360855854, 1.0: 0.01553038972325307, 2.0: 0.009729741230773657, 3.0: 0.015552544648116201,
           4.0: 0.021196186202801115, 5.0: 0.02284319986706472, 7.0: 0.010111423894480876, 10.0: 0.00613267811172937,
           15.0: 0.0036969930706003337, 20.0: 0.003791311459217732}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_controller(n_main, n_pre, backend):
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


def cpu_baseline(target_seconds=12.0):
    """the CPU oracle (oracle/mcx_oracle.c, OpenMP over paths) on a bounded sample of the same workload"""
    from oracle_backend import OracleBackend
    be = OracleBackend()
    threads = int(be.lib.orc_num_threads())

    def run(n):
        sc = build_controller(n, 16384, be)
        sc.prepare()
        t0 = time.perf_counter()
        sc.main_pass()
        return time.perf_counter() - t0, sc.sim_plan.n_steps

    t, S = run(16384)
    rate = 16384 * S / t
    n = int(min(1 << 20, max(16384, rate * target_seconds / S)))
    n = (n // 4096) * 4096
    t, S = run(n)
    return {"value": n * S / t, "unit": "path-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} paths x {S} steps of the same workload (K1+K2+K4 on the CPU oracle, OpenMP {threads} threads), {t:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths", type=int, default=1 << 20, help="main-simulation paths PER GPU")
    ap.add_argument("--presim", type=int, default=131072, help="pre-simulation (LSM) paths PER GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plan", default="auto", choices=["auto", "semi", "fused", "unfused"],
                    help="main-pass execution plan: semi = K1 + one book/metric kernel; fused = one launch; unfused = K1,K2,K4")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    from mcx import _native
    be = _native.HipBackend(local_rank)
    sc = build_controller(args.paths * world, args.presim * world, be)
    t0 = time.perf_counter()
    sc.prepare()
    be.synchronize()
    t_prepare = time.perf_counter() - t0
    S = sc.sim_plan.n_steps
    n_local = sc._main_engine.num_paths
    T, D, E = sc.sim_plan.n_dates, sc.sim_plan.n_state, len(sc.exposure_timeline)
    paths_buf = be.empty(T, D, n_local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    # two execution plans exist for the main pass (one fused launch / K1+K2+K4 launches); time both once, keep the faster
    plan_ms = {}
    fused_obj = sc._fused
    names = ["semi", "fused", "unfused"]
    if args.plan != "auto":
        names = [args.plan]

    def set_plan(name):
        sc._fused = None if name == "unfused" else fused_obj
        sc.main_plan = name if name != "unfused" else sc.main_plan

    for name in names:
        if name != "unfused" and fused_obj is None:
            continue
        set_plan(name)
        sc.main_pass(paths_buf if name != "fused" else None)
        barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            sc.main_pass(paths_buf if name != "fused" else None)
        barrier()
        plan_ms[name] = (time.perf_counter() - t0) / 2 * 1e3
    best = min(plan_ms, key=plan_ms.get)
    if world > 1:      # all ranks must agree
        flag = torch.tensor([float(names.index(best))], device="cuda")
        dist.broadcast(flag, 0)
        best = names[int(flag.item())]
    set_plan(best)
    for _ in range(args.warmup):
        res = sc.main_pass(paths_buf if best != "fused" else None)
    # per-kernel device time of the dominant kernel (K1) with HIP events on the launch stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    fused = best == "fused"
    for k in range(args.steps):
        ev[k][0].record()
        if fused:
            res = sc._fused_pass()               # one launch: K1+K2+K4 (+ block merge, record copy, rank gather)
            ev[k][1].record()
        elif best == "semi":
            paths = sc._main_engine.generate_paths_native(out=paths_buf)
            ev[k][1].record()                    # K1 device time; then ONE kernel for book + metrics
            rec = be.fused_eval_paths(sc._fused, paths)
            res = sc._finish_fused_records(rec)
        else:
            paths = sc._main_engine.generate_paths_native(out=paths_buf)
            ev[k][1].record()
            cfs, expo = be.eval_book(sc.book, paths)
            res = sc._evaluate_all(sc._shard, cfs, expo, paths)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    k1_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank == 0:
        cva, err = res[0][0][0]
        total_paths = n_local * world if world == 1 else args.paths * world
        value = total_paths * S * args.steps / dt
        pass_bytes = 8.0 * (2 * T * D + 2 * E + 2) * n_local   # SURVEY.md §8d B_path for the whole pass
        # HBM bytes per launch of the dominant kernel from the PMC counters (tools/measure_traffic.sh: separate FETCH_SIZE /
        # WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), measured at 2^20 paths
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = "kf_fused" if fused else "k1_paths"
            if key in pm:
                traffic = pm[key]["hbm_bytes_per_launch_at_1Mi_paths"] * (n_local / float(1 << 20))
        except Exception:
            traffic = None
        # dominant kernel: the fused pass carries the whole pass's algorithmic bytes; unfused K1 only its output tensor
        k1_bytes = pass_bytes if fused else 8.0 * T * D * n_local
        achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
        # the kernels are bound by f64 VALU issue, not HBM: companion figure from the SQ counters of
        # tools/measure_sq.sh (VALU instructions per wave at 2^20 paths) and the live kernel time
        alu = None
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))["kf_fused" if fused else "k1_paths"]
            cycles = sq["valu_insts_per_wave"] * sq["waves"] * (n_local / float(1 << 20)) * 4.0 / sq["n_simd"]
            alu = {"bound": "f64 VALU issue (4 clk per wave64 instruction)", "valu_insts_per_path_step": sq["valu_insts_per_wave"] * sq["waves"] * 64.0 / ((1 << 20) * S),
                   "valu_busy_ms_at_2.4GHz": cycles / 2.4e6, "frac": cycles / 2.4e6 / k1_ms}
        except Exception:
            alu = None
        out = {
            "metric": "path-steps/sec at 1M paths x 250 steps; PV/CVA rel-error vs CPU ref",
            "value": value, "unit": "path-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Vasicek+CIR++ (rho=0.5) payer IRS CVA, Euler, 51 dates x 5 sub-steps (SURVEY §8d config 3)",
                       "paths_per_gpu": n_local, "steps_per_path": S, "state_dim": D, "stored_dates": T,
                       "exposure_dates": E, "presim_paths_per_gpu": args.presim, "parallelism": f"paths x{world}",
                       "execution_plan": best, "plan_probe_ms": plan_ms},
            "roofline": {"bound": "hbm", "kernel": "kf_fused_lean<2,2,SIG_VAS_CIR_E> (Philox + Box-Muller + SDE + cashflows + LSM exposure + CVA in one launch)" if fused else "k1_paths<2,2,SIG_VAS_CIR_E> (Philox4x32-10 + Box-Muller + Cholesky + Vasicek/CIR++ Euler)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": k1_ms, "algorithmic_bytes_per_launch": k1_bytes,
                         "whole_pass_algorithmic_GBs": pass_bytes / (dt / args.steps) / 1e9},
            "alu": alu,
            "result": {"cva": cva, "mc_error": err},
            "prepare_s": t_prepare,
        }
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline()
            out["cpu_baseline"] = cb
            # PV/CVA vs the CPU reference path: same Philox stream at the sample size -> direct comparison below
            out["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
