#!/usr/bin/env python3
"""bench.py -- banded pair-HMM forward/backward/posterior throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step is one pass of the hot path (forward, checkpointed backward, posterior decode) over one
synthetic batch per GPU: BASELINE.json configs[2], 1024 reads x (10k events x 5k k-mers), band
(diagonalExpansion) 100, per-read scaled pore models.  Inputs are resident in HBM before the timed
region.  Reads shard across ranks with no data-path collective (weak scaling: every GPU gets its
own 1024 reads).  Consecutive steps are pipelined (--inflight 2): step s+1 works on another batch of 1024
reads on a stream of its own and is issued before step s is waited for, so two passes overlap on the GPU the
way queued batches do in service; the timed region still covers exactly K complete steps.  Rank 0 prints ONE
JSON line.

value      = in-band cells (each counted once) of all ranks / max-over-ranks wall time, Gcells/s
roofline   = algorithmic bytes (48 B per cell: one fp64 write + one re-read of 3 states,
             SURVEY.md section 8d) / average kernel duration measured with HIP events on the
             library's own stream, against the 8 TB/s HBM peak
cpu_baseline = the CPU oracle (a port of the reference's algorithm; the reference itself cannot be
             built here) timed on rank 0 over the first reads of the same batch: on all host cores of
             the job (a pool of forked workers, started before the process touches the GPU) and on
             one thread.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


_CPU = {}


def _cpu_read(i):
    """one read of the CPU baseline (oracle; worker of the all-core pool): returns its in-band cells"""
    import pyoracle as o
    batch, band = _CPU["batch"], _CPU["band"]
    it = batch["items"][i]
    m, gx, gy = batch["models"][it["model"]]
    x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
    ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
    an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
    p = o.default_params(threshold=0.01, minDiagsBetweenTraceBack=1000, traceBackDiagonals=40,
                         diagonalExpansion=band, splitMatrixBiggerThanThis=1 << 60)
    return o.aligned_pairs_using_anchors(o.Sm3Model(m, gy, gx), x, it["lX"], ev, an, p, True, True)["cells"]


def cpu_all_cores(batch, band, per_core, cores):
    """The CPU baseline on every host core: a pool of forked workers, one oracle call per read.  Runs BEFORE the
    process touches the GPU (forked children of a GPU-initialised process are not something to rely on)."""
    import multiprocessing as mp
    n = min(len(batch["items"]), per_core * cores)
    _CPU["batch"], _CPU["band"] = batch, band
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_read, range(min(cores, n)))  # load the library in every worker, untimed
        t0 = time.perf_counter()
        cells = sum(pool.map(_cpu_read, range(n), chunksize=1))
        dt = time.perf_counter() - t0
    return {"value": round(cells / dt / 1e9, 6), "unit": "Gcells/s", "cores": cores, "kind": "port",
            "seconds": round(dt, 2), "cells": int(cells),
            "sample": "first %d reads of the same batch, %d worker processes, oracle/cpecan_oracle.c" % (n, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1024)
    ap.add_argument("--events", type=int, default=10000)
    ap.add_argument("--kmers", type=int, default=5000)
    ap.add_argument("--band", type=int, default=100)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 general, 2 systolic")
    ap.add_argument("--cpu-reads", type=int, default=60,
                    help="reads timed on the CPU oracle, ~0.3 s each on one core (0: skip)")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="worker processes of the all-core CPU baseline (0: the host cores this job may use, at most 16)")
    ap.add_argument("--check", type=int, default=2, help="reads compared with the oracle after the run")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight: step s+1 (another batch, its own stream) is issued before step s is "
                         "waited for, as a server with queued batches would; every step is still one pass over "
                         "one batch and all K steps complete inside the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)

    import torch
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import synth
    # the all-core CPU baseline runs first: before the HIP library is loaded and a context exists
    cpu_all = None
    first_batch = None
    if rank == 0 and world == 1 and args.cpu_reads > 0:
        first_batch = synth.make_batch(3, args.reads, args.kmers, args.events, anchor_every=50)
        # a one-GPU box of the pool gives a job 16 host cores, whatever os.cpu_count() says about the machine
        cores = args.cpu_cores or min(len(os.sched_getaffinity(0)), 16)
        cpu_all = cpu_all_cores(first_batch, args.band, max(1, args.cpu_reads * 2 // 3), cores)
    from cpecan_load import binding
    cp = binding()

    # ---- inputs: this rank's reads, uploaded before the timed region -------------------------
    # `inflight` batches of distinct reads, each with a context (stream) of its own.  With two in flight every
    # batch runs as ONE stream group (two kernels of 1024 workgroups overlap); alone, a batch is split into two
    # groups so that its own forward and backward kernels overlap.
    inflight = max(1, min(args.inflight, args.steps))
    if inflight > 1 and "CPECAN_SYSTOLIC_GROUPS" not in os.environ:
        os.environ["CPECAN_SYSTOLIC_GROUPS"] = "1"
    bp = cp.BandParams(0.01, 1000, 40, args.band)
    t_gen = t_models = t_upload = 0.0
    batches, ctxs, bs = [], [], []
    for j in range(inflight):
        t0 = time.time()
        config_id = 3 + 100 * rank + 10 * j  # distinct read seeds per rank and per batch in flight
        bt = first_batch if (j == 0 and first_batch is not None) else \
            synth.make_batch(config_id, args.reads, args.kmers, args.events, anchor_every=50)
        t_gen += time.time() - t0
        cx = cp.Context(local_rank)
        t0 = time.time()
        cx.models_create([(cp.NANOPORE_TRANSITIONS, m, gx, gy) for (m, gx, gy) in bt["models"]])
        t_models += time.time() - t0
        items = np.zeros(len(bt["items"]), cp.ITEM_DTYPE)
        for i, it in enumerate(bt["items"]):
            items[i] = (it["x_offset"], it["lX"], it["y_offset"], it["lY"], it["anchor_offset"],
                        it["n_anchors"], it["model"], 1, 1, 0)  # ragged ends, as vanillaAlign.c:203
        t0 = time.time()
        bs.append(cp.Batch(cx, items, bt["x_chars"], bt["events"], bt["anchors"], bp,
                           cp.MODE_POSTERIOR, args.kernel, 0))
        t_upload += time.time() - t0
        batches.append(bt)
        ctxs.append(cx)
    batch, b = batches[0], bs[0]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    kernel_ms, stage = [], []

    def run_steps(n, record):
        """n steps, step s on batch s % inflight; a batch is waited for only when it is needed again"""
        pending = [False] * inflight
        for s_ in range(n):
            j = s_ % inflight
            if pending[j]:
                bs[j].sync()
                if record:
                    kernel_ms.append(bs[j].elapsed_ms()[1])
                    if bs[j].info()["kernel"] == "systolic":
                        stage.append(bs[j].stage_ms())
            bs[j].run()
            pending[j] = True
        for j in range(inflight):
            if pending[j]:
                bs[j].sync()
                if record:
                    kernel_ms.append(bs[j].elapsed_ms()[1])
                    if bs[j].info()["kernel"] == "systolic":
                        stage.append(bs[j].stage_ms())

    run_steps(max(args.warmup, inflight if args.warmup > 0 else 0), False)
    sync_all()
    t_start = time.perf_counter()
    run_steps(args.steps, True)
    sync_all()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    npairs, ntot, ncells = b.counts()
    cells_of = [int(x.counts()[2].sum()) for x in bs]
    cells = int(round(sum(cells_of[s_ % inflight] for s_ in range(args.steps)) / args.steps))  # per step
    if world > 1:
        ct = torch.tensor([cells], dtype=torch.int64, device="cuda")
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        total_cells = int(ct.item())
    else:
        total_cells = cells

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    gcells = total_cells * args.steps / elapsed / 1e9
    reads_per_s = args.reads * world * args.steps / elapsed
    # per-step time of this rank: with batches in flight the steps overlap, so it is the rank's wall time over
    # its steps (HIP-event times of single runs overlap one another)
    avg_kernel_s = elapsed / args.steps if inflight > 1 else float(np.mean(kernel_ms)) / 1e3
    bytes_per_cell = 48.0
    achieved = cells * bytes_per_cell / avg_kernel_s / 1e9
    # HBM traffic of one pass from the PMC counters (FETCH_SIZE / WRITE_SIZE collected in separate
    # rocprofv3 --pmc passes of this same command; summary committed under profiles/)
    traffic, traffic_note, valu = None, None, None
    prof = os.path.join(ROOT, "profiles", "r01_rocprofv3_pmc_summary.json")
    if os.path.exists(prof) and args.reads == 1024 and args.events == 10000 and args.kmers == 5000:
        pm = json.load(open(prof))
        # FETCH_SIZE reports half the bytes of coalesced streaming reads on gfx950 (the guide's rule; the
        # calibration kernels in the same profile confirm it for this path's 8-byte-per-lane loads:
        # reported/true = 0.5); WRITE_SIZE is exact
        rd = 2.0 * pm["FETCH_SIZE_GB_per_pass"] * 1e9
        wr = pm["WRITE_SIZE_GB_per_pass"] * 1e9
        traffic = rd + wr
        traffic_note = ("HBM bytes per pass over the batch from rocprofv3 --pmc (separate passes): "
                        "2 x FETCH_SIZE = %.1f GB + WRITE_SIZE = %.1f GB; algorithmic %.1f GB"
                        % (rd / 1e9, wr / 1e9, cells * 48.0 / 1e9))
        valu = {k: round(v.get("valu_busy_fraction_of_simd_time", 0.0), 3) for k, v in pm.get("SQ", {}).items()}
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_note": traffic_note,
                "scope": "whole pass (all kernels of the path), 48 B per cell",
                "kernel_ms": round(1e3 * avg_kernel_s, 3), "bytes_per_cell": bytes_per_cell,
                "frac_of_measured_copy_6290": round(achieved / 6290.0, 5),
                "fp64_valu_busy": valu,
                "note": "fp64 log-space recurrence, not HBM-bound: fp64_valu_busy is the fraction of SIMD time "
                        "with a VALU instruction executing (committed PMC profile); the rest is the dependency "
                        "chain of an anti-diagonal (barrier, LDS exchange, serial log-adds) at 4 workgroups per "
                        "CU -- DESIGN.md section 5"}
    if stage:
        f_ms = float(np.mean([x[0] for x in stage]))
        k_ms = float(np.mean([x[1] for x in stage]))
        n_l = stage[0][2]
        sfx = "_r3" if b.info().get("waves_per_workgroup") == 3 else ""  # the three-wave build of the kernels
        # dominant kernel: the backward-window kernel re-reads the 3 forward states of every cell
        # once (24 B per cell); the forward-window kernel writes them once (24 B per cell)
        roofline["dominant_kernel"] = {
            "name": "cpecan_k_sy_backward" + sfx, "launches_per_pass": n_l,
            "note": "launches of the batches in flight (and of a batch's stream groups) overlap, so the sum of "
                    "launch durations exceeds the pass time",
            "avg_launch_ms": round(k_ms / n_l, 4),
            "algorithmic_bytes_per_launch": round(cells * 24.0 / n_l),
            "achieved": round(cells * 24.0 / (k_ms / 1e3) / 1e9, 2),
            "frac": round(cells * 24.0 / (k_ms / 1e3) / 1e9 / 8000.0, 5)}
        roofline["forward_kernel"] = {
            "name": "cpecan_k_sy_forward" + sfx, "launches_per_pass": n_l,
            "avg_launch_ms": round(f_ms / n_l, 4),
            "achieved": round(cells * 24.0 / (f_ms / 1e3) / 1e9, 2),
            "frac": round(cells * 24.0 / (f_ms / 1e3) / 1e9 / 8000.0, 5)}

    if os.environ.get("CPECAN_PROF"):  # timing build (-DSY_PROFILE) only
        import ctypes
        buf = (ctypes.c_ulonglong * 80)()
        if hasattr(cp.lib(), "cpecan_systolic_prof_fetch") and cp.lib().cpecan_systolic_prof_fetch(buf) == 0:
            n = max(buf[72], 1)
            sys.stderr.write("prof backward: %d windows, cycles per window: sweep %.0f totals %.0f | decode: pass0 %.0f sync %.0f prefix %.0f pass1 %.0f sync %.0f tail %.0f\n"
                             % tuple([buf[72]] + [buf[64 + k] / n for k in range(8)]))
            sys.stderr.write("prof backward: windows decoded by the scan %d, candidates per window %.0f\n" % (buf[73], buf[74] / n))
            sys.stderr.write("prof backward: slowest window %d cycles (max over all windows; the device counter is cumulative over launches)\n" % buf[75])
            sys.stderr.write("prof backward: window wall time (100 MHz ticks): mean %.0f max %d\n" % (buf[76] / n, buf[77]))
            for w in range(4):
                v = [buf[w * 16 + k] for k in range(12)]
                na, ni = max(v[10], 1), max(v[11], 1)
                sys.stderr.write("prof wave %d: active steps %d  cycles/step by section %s | inactive steps %d cycles/step %.0f\n"
                                 % (w, v[10], " ".join("%.0f" % (x / na) for x in v[:9]), v[11], v[9] / ni))
    # ---- parity spot check + CPU baseline (oracle = checker / baseline only) -------------------
    check = {"reads": 0}
    cpu = None
    cpu_reads = args.cpu_reads if world == 1 else 0  # the CPU baseline is a one-GPU (N=1) figure
    if args.check > 0 or cpu_reads > 0:
        from harness import assert_same_pairs, run_oracle_item
        n_cpu = max(args.check, cpu_reads)
        t_cpu, cpu_cells = 0.0, 0
        for i in range(min(n_cpu, args.reads)):
            t0 = time.perf_counter()
            ref = run_oracle_item(batch, i, bp, (1, 1))
            t_cpu += time.perf_counter() - t0
            cpu_cells += ref["cells"]
            if i < args.check:
                tri, lp = b.pairs(i, npairs[i])
                xay, tot = b.totals(i, ntot[i])
                assert_same_pairs(dict(triples=tri, logp=lp), ref)
                assert np.array_equal(tot, ref["totals"])
                check["reads"] += 1
        if cpu_reads > 0:
            one = {"value": round(cpu_cells / t_cpu / 1e9, 6), "unit": "Gcells/s", "cores": 1,
                   "kind": "port", "seconds": round(t_cpu, 2),
                   "sample": "first %d reads of the same batch, one thread, oracle/cpecan_oracle.c"
                             % min(n_cpu, args.reads)}
            cpu = dict(cpu_all, single_core=one) if cpu_all else one

    out = {
        "metric": "banded fwd-bwd Gcells/s", "value": round(gcells, 4), "unit": "Gcells/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "reads_per_s": round(reads_per_s, 1),
        "config": {"workload": "BASELINE configs[2]: %d reads/GPU x (%d events x %d k-mers), "
                               "diagonalExpansion %d, 3-state strawMan signal HMM, per-read scaled "
                               "models, posterior decode" % (args.reads, args.events, args.kmers, args.band),
                   "cells_per_gpu": cells, "pairs_per_gpu": int(npairs.sum()),
                   "kernel": b.info(),
                   "batches_in_flight": inflight,
                   "step_latency_ms": round(float(np.mean(kernel_ms)), 3),
                   "parallelism": "reads sharded over %d GPU(s), no collective" % world},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity_check": check,
        "host_prep_s": {"generate": round(t_gen, 2), "derive_models": round(t_models, 2),
                        "upload_and_band": round(t_upload, 2)},
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
