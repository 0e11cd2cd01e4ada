#!/usr/bin/env python3
"""bench.py -- trajectory frames/s for full_analysis (pore + windows) on the
1000-frame synthetic CC3 trajectory (BASELINE.json config 2), N GPUs of one node.

One *step* = one pass of the whole hot path (all stages) over the rank's batch
of 1000 frames, inputs already resident in HBM.  Weak scaling: every rank
analyses its own 1000 frames (frames rank*1000 .. rank*1000+999 of the synthetic
generator); no data-path collective.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pipeline keeps several kernels in flight on separate streams: ask the HIP runtime for enough
# hardware queues BEFORE anything (torch included) initialises it
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FRAMES = 1000
ALGO_BYTES_PER_UNIT = 24 * 168 + 600   # coordinates read once + one result record
ALGO_FLOP_PER_UNIT = 2.0e7
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01j_hbm_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary
FP64_VECTOR_PEAK_TFLOPS = 78.6
COUNTER_FILE = os.path.join(ROOT, "profiles", "r01j_instruction_counters.json")    # rocprofv3 --pmc SQ_INSTS_* summary


def cpu_baseline(elements, frames, vdw, mass, budget_s=20.0):
    """The oracle (numpy/scipy/sklearn restatement of the reference's path,
    bit-identical to it on the golden inputs) timed on this host, one core."""
    from oracle import pw_oracle as O

    O.build()
    t0 = time.perf_counter()
    n = 0
    while n < len(frames) and (n < 2 or time.perf_counter() - t0 < budget_s):
        O.full_analysis(frames[n], vdw, mass)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"first {n} frames of the same synthetic trajectory, oracle/pw_oracle.py, 1 process"}


def secondary(ctx, elements, vdw, mass):
    """BASELINE.json's secondary shapes, reported beside the headline (not part of `value`):
    a large throughput batch (configs 4-5: tens of thousands of independent units) and the
    periodic pipeline of configs 3-4 (cell -> rebuilt cages -> analysis)."""
    from pywindow_amd import _lib, synth
    from pywindow_amd import rebuild as rb

    out = {}
    _, big = synth.synthetic_units(4000, first=50000)
    res = ctx.upload(_lib.Batch.uniform(big, vdw, mass))
    ms = res.time_launches(3)
    res.free()
    out["throughput_batch"] = {"units": 4000, "ms_per_launch": ms, "units_per_s": 4000 / (ms * 1e-3)}
    cell = os.path.join(ROOT, "tests", "golden", "rebuild.npz")
    if os.path.exists(cell):
        g = np.load(cell)
        el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
        frames = 256
        rng = np.random.default_rng(4)
        coords = xyz[None] + rng.normal(0.0, 0.02, size=(frames,) + xyz.shape)
        topo = rb.CellTopology(el)
        lats = np.repeat(lat[None], frames, axis=0)
        from pywindow_amd import element_data as E

        ids = E.element_ids(el)
        cc, ll, inv = rb.pack_frames(coords, lats)

        def run():
            res, n_mol = ctx.resident_from_cells(topo, E.VDW[ids], cc, ll, inv, True)
            res.launch()
            recs = res.download()
            res.free()
            return n_mol, recs

        run()
        t0 = time.perf_counter()
        n_mol, recs = run()
        dt = time.perf_counter() - t0
        out["periodic_cell"] = {
            "workload": "cubic cell, 8 CC3 cages / 1344 atoms per frame (tests/data/system_periodic.pdb + 0.02 A noise)",
            "frames": frames, "cages": int(n_mol.sum()), "ms": 1e3 * dt,
            "frames_per_s": frames / dt, "cages_per_s": float(n_mol.sum()) / dt,
            "includes": "H2D of the frames, rebuild launch, on-device hand-over, analysis launch, D2H of the records",
            "all_cages_have_windows": bool((recs["n_windows"] > 0).all())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch

    dist = None
    backend = os.environ.get("PW_BENCH_BACKEND", "nccl")       # "gloo": rehearsal of the multi-rank path
    if "PW_BENCH_DEVICE" in os.environ:                        # ... with every rank on one GPU
        local_rank = int(os.environ["PW_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(args.frames, first=rank * args.frames)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    ctx = _lib.Context(local_rank)
    res = ctx.upload(_lib.Batch.uniform(frames, vdw, mass))

    def barrier():
        res.sync()                     # the engine's own HIP streams
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        res.launch()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res.launch()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = res.download()
    ok = bool((out["status"] == 0).all())

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * args.frames * args.steps / elapsed
        # kernel duration measured with HIP events on the launch stream
        k_ms = res.time_launches(max(3, min(args.steps, 10)))
        # one launch on its own (no overlap with a neighbour): the latency of a single batch
        lat = []
        for _ in range(3):
            res.sync()
            t1 = time.perf_counter()
            res.launch()
            res.sync()
            lat.append(1e3 * (time.perf_counter() - t1))
        single_ms = min(lat)
        units_per_s = args.frames / (k_ms * 1e-3)
        achieved_gbs = units_per_s * ALGO_BYTES_PER_UNIT / 1e9
        traffic = None
        try:
            with open(TRAFFIC_FILE) as fh:
                tj = json.load(fh)
            traffic = tj["per_launch_bytes"] * (args.frames / tj["units_per_launch"])
        except (OSError, KeyError, ValueError):
            traffic = None
        # VALU issue utilisation from the profiled wave-level instruction count (committed summary) and
        # the kernel time measured in this run: instructions x 4 SIMD cycles / (SIMDs x clock x time)
        valu_issue = None
        try:
            with open(COUNTER_FILE) as fh:
                cj = json.load(fh)
            per_unit = cj["per_launch"]["SQ_INSTS_VALU"] / cj["units_per_launch"]
            vi = cj["valu_issue"]
            valu_issue = {"wave_instructions_per_launch": per_unit * args.frames,
                          "frac": per_unit * args.frames * vi["simd_cycles_per_wave_instruction"]
                                  / (vi["simds"] * vi["clock_ghz"] * 1e9 * k_ms * 1e-3),
                          "source": "profiles/r01j_instruction_counters.json (rocprofv3 --pmc SQ_INSTS_VALU)"}
        except (OSError, KeyError, ValueError):
            valu_issue = None
        line = {
            "metric": "trajectory frames/sec full_analysis (pore+windows), CC3 1k-frame",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "CC3 1000-frame synthetic DL_POLY trajectory (BASELINE configs[1]), "
                                   "per-frame pore+windows, 168 atoms/frame",
                       "frames_per_gpu": args.frames, "stages": "all", "results_ok": ok,
                       "successive_steps_overlap": True, "single_step_latency_ms": single_ms,
                       "windows_eq_4": int((out["n_windows"] == 4).sum())},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "fabric bytes per launch from rocprofv3 PMC (profiles/r01j_hbm_traffic.json); "
                                         "includes Infinity-Cache hits on the re-used per-team workspaces",
                         "kernel": "pw_analyse_kernel (pipeline: optimiser chains | average diameter | window search)", "kernel_ms": k_ms,
                         "kernel_ms_note": "HIP events around back-to-back analyses / their number: the steady-state period of "
                                           "the pipeline, whose three launches per analysis overlap each other and the next "
                                           "analysis; rocprofv3 therefore reports longer per-launch durations "
                                           "(profiles/r01j_pipeline_kernel_stats.csv: chains 3.8 ms, window search 3.6 ms, "
                                           "average diameter 1.8 ms, all inside config.single_step_latency_ms)",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_UNIT * args.frames,
                         "fp64_valu": {"achieved_tflops": units_per_s * ALGO_FLOP_PER_UNIT / 1e12,
                                       "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                       "frac": units_per_s * ALGO_FLOP_PER_UNIT / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                       "valu_issue_measured": valu_issue}},
        }
        if not args.no_secondary and world == 1:
            line["secondary"] = secondary(ctx, elements, vdw, mass)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(elements, frames, vdw, mass)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
