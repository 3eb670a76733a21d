#!/usr/bin/env python3
"""bench.py -- banded pair-HMM forward/backward/posterior throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode posterior|em]

N > 1 without a rendezvous in the environment: this process starts its own N workers (one per GPU, through
`python -m torch.distributed.run` on 127.0.0.1), before it touches a GPU itself, and passes their output on.  The
driver's own `torch.distributed.run ... bench.py --gpus N` launch is recognised by WORLD_SIZE and runs as a worker.

--mode posterior (default).  A step is one pass of the hot path (forward, checkpointed backward, posterior decode,
the aligned pairs finished on the host) over one synthetic batch per GPU: BASELINE.json configs[2], 1024 reads x
(10k events x 5k k-mers), band (diagonalExpansion) 100, per-read scaled pore models.  Inputs are resident in HBM
before the timed region.  Reads shard across ranks with no data-path collective (weak scaling: every GPU gets its
own 1024 reads).  Consecutive steps are pipelined (--inflight 2): step s+1 works on another batch of 1024 reads on
streams of its own and is issued before step s is waited for, the way queued batches overlap in service; while the
GPU works on one batch the host finishes the other's pairs (exp, threshold, floor with libm -- what makes the
integer posteriors the reference's, bit for bit).  The timed region covers exactly K complete steps, pairs included.
After it, rank 0 times ONE batch alone (--single-steps, each step waited for before the next is issued) on the
wave-per-alignment kernels: roofline.frac_single_batch.

--mode em.  BASELINE.json configs[3]: a step is one Baum-Welch iteration -- E-step of this rank's reads on the GPU
(expectations summed on the device), ONE all-reduce of the [9 transitions | 4096 k-mer gaps | likelihood] vector
over RCCL, M-step (normalise, reload the transitions and gap probabilities in place on the device).

value      = in-band cells (each counted once) of all ranks / max-over-ranks wall time, Gcells/s
roofline   = algorithmic bytes (48 B per cell: one fp64 write + one re-read of 3 states) /
             average duration of a pass, against the 8 TB/s HBM peak; dominant_kernel from HIP events on the
             library's own streams
cpu_baseline = the CPU oracle (a port of the reference's algorithm; the reference itself cannot be built here:
             sonLib is absent) timed on rank 0 over the first reads of the same batch: on all host cores of the job
             (a pool of forked workers, started before the process touches the GPU) and on one thread.

--rehearse: the N-rank plumbing without a GPU (gloo; a step is the host band geometry of the rank's reads) -- what
the CPU tests run with two ranks; its JSON line says so and carries no throughput claim.
"""
import argparse
import json
import os
import subprocess
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("posterior", "em", "hdp", "service"), default="posterior")
    ap.add_argument("--config", type=int, default=0,
                    help="BASELINE.json configs by number: 3 = --mode posterior (configs[2]), 4 = --mode em (configs[3]), "
                         "5 = --mode hdp (configs[4])")
    ap.add_argument("--hdp", default=os.path.join(ROOT, "tests", "golden", "testTemplate.nhdp"),
                    help="--mode hdp: a serialized NanoporeHDP (default: the reference's own test fixture)")
    ap.add_argument("--reads", type=int, default=1024)
    ap.add_argument("--events", type=int, default=10000)
    ap.add_argument("--kmers", type=int, default=5000)
    ap.add_argument("--band", type=int, default=100)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 general, 2 register-resident")
    ap.add_argument("--family", choices=("auto", "wave", "workgroup"), default="auto",
                    help="register-resident kernels of the timed region: wave-per-alignment or workgroup-per-"
                         "alignment; auto = workgroup when batches are pipelined (--inflight 2), wave otherwise")
    ap.add_argument("--cpu-reads", type=int, default=60,
                    help="reads timed on the CPU oracle, ~0.3 s each on one core (0: skip)")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="worker processes of the all-core CPU baseline (0: the host cores this job may use, at most 16)")
    ap.add_argument("--check", type=int, default=2, help="reads compared with the oracle after the run")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight: step s+1 (another batch, its own streams) is issued before step s is "
                         "waited for, as a server with queued batches would; every step is still one pass over "
                         "one batch and all K steps complete inside the timed region")
    ap.add_argument("--single-steps", type=int, default=12,
                    help="steps of the single-batch measurement after the timed region (rank 0, N=1; 0: skip)")
    ap.add_argument("--no-finalise", action="store_true",
                    help="leave the pairs as (x, y, exponent) in HBM inside the timed region instead of finishing "
                         "them on the host (GPU pass alone)")
    ap.add_argument("--host-tables", action="store_true",
                    help="--mode service: every read's scaled model table comes from the caller (cpecan_hip_models_create) "
                         "instead of one pore model + per-read scaling parameters (cpecan_hip_models_create_scaled)")
    ap.add_argument("--service-slots", type=int, default=4,
                    help="--mode service: batches alive at once (contexts): running, queued, being prepared")
    ap.add_argument("--service-threads", type=int, default=2,
                    help="--mode service: host threads that prepare and queue batches")
    ap.add_argument("--em-contexts", type=int, default=4,
                    help="--mode em: 2..4 = the rank's reads as that many concurrent batches on the workgroup kernels "
                         "(4: 105 ms per iteration for 1024 C3 reads, 2: 110), 1 = one batch on the wave kernels (150)")
    ap.add_argument("--rehearse", action="store_true", help="no GPU: gloo ranks, host band geometry as the step")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-started workers")
    args = ap.parse_args(argv)
    if args.config:
        args.mode = {3: "posterior", 4: "em", 5: "hdp"}.get(args.config) or sys.exit("--config takes 3, 4 or 5")
    given = argv if argv is not None else sys.argv
    if args.mode == "hdp":  # long reads: its own default sizes
        if "--reads" not in given:
            args.reads, args.events, args.kmers = 1024, 50000, 41500
        if "--steps" not in given:
            args.steps, args.warmup = 5, 1
    return args


def launch_workers(args, argv):
    """--gpus N from a plain command line: N worker processes, one per GPU, started before this process touches a
    GPU; their stdout (rank 0's JSON line) is passed through, the exit code is theirs."""
    port = args.master_port or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def make_items(cp, bt):
    items = np.zeros(len(bt["items"]), cp.ITEM_DTYPE)
    for i, it in enumerate(bt["items"]):
        items[i] = (it["x_offset"], it["lX"], it["y_offset"], it["lY"], it["anchor_offset"],
                    it["n_anchors"], it["model"], 1, 1, 0)  # ragged ends, as vanillaAlign.c:203
    return items


def rehearse(args, rank, world):
    """The ranks' plumbing on the CPU: rendezvous, shard by rank, barrier, max-over-ranks time, sum of cells, one JSON
    line from rank 0.  The step is the host part of a pass that needs no GPU: the band table of every read."""
    import torch
    import torch.distributed as dist
    import synth
    from cpecan_load import binding
    cp = binding()
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    bt = synth.make_batch(3 + 100 * rank, args.reads, args.kmers, args.events, anchor_every=50)

    def step():
        cells = 0
        for it in bt["items"]:
            an = bt["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
            lo, hi = cp.band_construct(an, it["lX"], it["lY"], args.band)
            cells += int(((hi - lo) // 2 + 1).sum())
        return cells

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    cells = 0
    for _ in range(args.steps):
        cells = step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64)
    ct = torch.tensor([cells], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({
            "metric": "banded fwd-bwd Gcells/s", "value": None, "unit": "Gcells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * float(tt.item()) / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic; REHEARSAL on the CPU (gloo): host band geometry only, no GPU work, no throughput claim",
            "config": {"workload": "%d reads/rank x (%d events x %d k-mers), diagonalExpansion %d"
                                   % (args.reads, args.events, args.kmers, args.band),
                       "cells_all_ranks": int(ct.item()), "cells_rank0": cells}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_workers(args, argv)
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse:
        return rehearse(args, rank, world)

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
    from cpecan_load import binding, em as load_em
    cp = binding()
    bp = cp.BandParams(0.01, 1000, 40, args.band)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == "service":
        return bench_service(args, cp, bp, rank, local_rank, world, synth)
    if args.mode == "hdp":
        return bench_hdp(args, cp, bp, rank, local_rank, world, dist, torch, sync_all)
    if args.mode == "em":
        return bench_em(args, cp, load_em(), bp, rank, local_rank, world, dist, torch, synth, sync_all, first_batch,
                        cpu_all)

    # ---- inputs: this rank's reads, uploaded before the timed region -------------------------
    # `inflight` batches of distinct reads, each with a context (streams) of its own.  Pipelined batches run on the
    # workgroup-per-alignment kernels, each as ONE stream group (two kernels of 1024 workgroups overlap); alone, a
    # batch runs on the wave-per-alignment kernels, whose forward sweep of window w+1 overlaps the backward sweep of
    # window w inside the batch.
    inflight = max(1, min(args.inflight, args.steps))
    # auto: the wave-per-alignment kernels where the batch runs their hand-scheduled assembly sweeps (asked of the first
    # batch below) -- pipelined batches then follow one another's last forward sweep; otherwise as before: workgroup
    # kernels for pipelined batches, wave kernels for a batch alone
    family = args.family if args.family != "auto" else "wave"
    flags = cp.FLAG_WORKGROUP_KERNELS if family == "workgroup" else 0
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
        t0 = time.time()
        bs.append(cp.Batch(cx, make_items(cp, bt), bt["x_chars"], bt["events"], bt["anchors"], bp,
                           cp.MODE_POSTERIOR, args.kernel, flags))
        if j == 0 and args.family == "auto" and inflight > 1 and bs[0].info().get("assembly_sweeps", 0) != 2:
            family, flags = "workgroup", cp.FLAG_WORKGROUP_KERNELS
            os.environ.setdefault("CPECAN_SYSTOLIC_GROUPS", "1")
            bs[0].close()
            bs[0] = cp.Batch(cx, make_items(cp, bt), bt["x_chars"], bt["events"], bt["anchors"], bp,
                             cp.MODE_POSTERIOR, args.kernel, flags)
        t_upload += time.time() - t0
        batches.append(bt)
        ctxs.append(cx)
    batch, b = batches[0], bs[0]
    finalise = not args.no_finalise

    kernel_ms, stage, fin_s, clock_mhz = [], [], [], []
    chained = family == "wave"

    def wait_for(j, record):
        bs[j].sync()
        if record:
            kernel_ms.append(bs[j].elapsed_ms()[1])
            mhz = bs[j].shader_clock_mhz()  # from the sweeps' own counters (wave kernels): boxes differ under this load
            if mhz > 0:
                clock_mhz.append(mhz)
            if bs[j].info()["kernel"] == "systolic":
                stage.append(bs[j].stage_ms())
        if finalise:  # D2H of the candidates + exp/threshold/floor on the host threads; the GPU works on the other batch
            t0 = time.perf_counter()
            bs[j].counts()
            if record:
                fin_s.append(time.perf_counter() - t0)

    def run_steps(n, record):
        """n steps, step s on batch s % inflight; a batch is waited for only when it is needed again"""
        pending = [False] * inflight
        for s_ in range(n):
            j = s_ % inflight
            if pending[j]:
                wait_for(j, record)
            # wave kernels: one pass at a time on the device (a batch fills the register files); the batch's kernels
            # are ordered behind the previous batch's on the device, the host meanwhile finishes that one's pairs
            prev = bs[(s_ - 1) % inflight] if (chained and inflight > 1 and s_ > 0) else None
            bs[j].run(after=prev)
            pending[j] = True
        for k in range(inflight):  # the oldest first
            j = (n + k) % inflight
            if pending[j]:
                wait_for(j, record)

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

    t0 = time.perf_counter()
    npairs, ntot, ncells = b.counts()
    t_fetch = time.perf_counter() - t0  # nothing left to do when the timed region finished the pairs
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
        return 0

    ms_per_step = 1e3 * elapsed / args.steps
    gcells = total_cells * args.steps / elapsed / 1e9
    reads_per_s = args.reads * world * args.steps / elapsed
    # per-step time of this rank: with batches in flight the steps overlap, so it is the rank's wall time over
    # its steps (HIP-event times of single runs overlap one another)
    avg_pass_s = elapsed / args.steps if inflight > 1 else max(float(np.mean(kernel_ms)) / 1e3, 1e-9)
    bytes_per_cell = 48.0
    achieved = cells * bytes_per_cell / avg_pass_s / 1e9
    info = b.info()

    # ---- one batch alone (the figure the 40 % target is quoted on) ----------------------------
    single = None
    if args.single_steps > 0 and world == 1:
        for x in bs[1:]:
            x.close()
        if family == "workgroup":
            os.environ.pop("CPECAN_SYSTOLIC_GROUPS", None)
        single = {}
        for fam, fl in (("wave", 0), ("workgroup", cp.FLAG_WORKGROUP_KERNELS)):
            sb = cp.Batch(ctxs[0], make_items(cp, batch), batch["x_chars"], batch["events"], batch["anchors"], bp,
                          cp.MODE_POSTERIOR, args.kernel, fl)
            for _ in range(3):
                sb.run()
                sb.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev_ms, st = [], []
            for _ in range(args.single_steps):
                sb.run()
                sb.sync()
                ev_ms.append(sb.elapsed_ms()[1])
                if sb.info()["kernel"] == "systolic":
                    st.append(sb.stage_ms())
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / args.single_steps
            ms = float(np.mean(ev_ms))
            t0 = time.perf_counter()
            sb.counts()
            t_fin = time.perf_counter() - t0
            single[fam] = {"ms_per_pass": round(ms, 3), "wall_ms_per_pass": round(1e3 * wall, 3),
                           "gcells_per_s": round(cells_of[0] / (ms / 1e3) / 1e9, 3),
                           "frac": round(cells_of[0] * bytes_per_cell / (ms / 1e3) / 1e9 / 8000.0, 5),
                           "host_finalise_ms": round(1e3 * t_fin, 2), "kernel": sb.info(),
                           "stage": kernel_stage(sb.info(), st, cells_of[0])}
            mhz = sb.shader_clock_mhz()
            if mhz > 0:  # the sweeps' own cycle counters over their 100 MHz reference counters (wave kernels)
                single[fam]["shader_clock_mhz"] = round(mhz, 0)
            sb.close()

    # HBM traffic of one pass from the PMC counters (FETCH_SIZE / WRITE_SIZE collected in separate
    # rocprofv3 --pmc passes of this same command; summary committed under profiles/)
    traffic, traffic_note, valu = None, None, None
    asm = info.get("assembly_sweeps", 0) == 2
    prof = os.path.join(ROOT, "profiles", "r03_rocprofv3_pmc_summary.json" if asm else "r02_rocprofv3_pmc_summary.json")
    if os.path.exists(prof) and args.reads == 1024 and args.events == 10000 and args.kmers == 5000:
        pm = json.load(open(prof))["families"].get("assembly" if asm else family)
        if pm and pm.get("FETCH_SIZE_GB_per_pass"):
            # FETCH_SIZE reports half the bytes of coalesced streaming reads on gfx950 (the guide's rule; the
            # calibration kernels in the same profile confirm it for this path's 8-byte-per-lane loads:
            # reported/true = 0.5); WRITE_SIZE is exact
            rd = 2.0 * pm["FETCH_SIZE_GB_per_pass"] * 1e9
            wr = pm["WRITE_SIZE_GB_per_pass"] * 1e9
            traffic = rd + wr
            traffic_note = ("HBM bytes per pass over the batch, %s kernels, from rocprofv3 --pmc (separate passes, "
                            "profiles/%s): 2 x FETCH_SIZE = %.1f GB + WRITE_SIZE = %.1f GB; "
                            "algorithmic %.1f GB" % ("assembly sweeps, wave" if asm else family, os.path.basename(prof),
                                                     rd / 1e9, wr / 1e9, cells * 48.0 / 1e9))
            valu = {k: round(v.get("valu_busy_fraction_of_simd_time", 0.0), 3) for k, v in pm.get("SQ", {}).items()}
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_note": traffic_note,
                "scope": "whole pass (all kernels of the path), 48 B per cell, %d batch(es) in flight, %s kernels"
                         % (inflight, family),
                "pass_ms": round(1e3 * avg_pass_s, 3), "bytes_per_cell": bytes_per_cell,
                "frac_of_measured_copy_6290": round(achieved / 6290.0, 5),
                "fp64_valu_busy": valu,
                "note": ("fp64 log-space recurrence with the reference's approximate logAdd reproduced bit for bit.  The "
                         "assembly sweeps run at the rate HBM moves their traffic: forward writes and backward reads of "
                         "the ring overlap at about the rate of a copy (4.6-4.9 TB/s measured on these boxes); underneath, "
                         "each wave waits for its own previous vector instruction (8.8 cycles for a dependent one), which "
                         "is why the sweeps' arithmetic is emitted stage by stage over the three layers -- DESIGN.md "
                         "section 5")
                        if asm else
                        ("fp64 log-space recurrence with the reference's approximate logAdd reproduced bit for bit: "
                         "the compiled kernels are bound by VALU issue (one fp64 instruction per SIMD per ~4 cycles) -- "
                         "DESIGN.md section 5 gives the instruction count per cell and the time it implies")}
    if single:
        best = max(single, key=lambda k: single[k]["frac"])
        roofline["frac_single_batch"] = single[best]["frac"]
        roofline["single_batch"] = dict(single, best=best,
                                        note="ONE batch of %d reads, each pass waited for before the next is "
                                             "issued; HIP-event time of the pass" % args.reads)
    st = kernel_stage(info, stage, cells)
    if st:
        roofline.update(st)

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

    # from host buffers to host-visible pairs for one batch: table derivation + upload/band + pass + pairs
    prep = (t_models + t_upload) / inflight
    e2e = prep + ms_per_step / 1e3 + (0.0 if finalise else t_fetch)
    out = {
        "metric": "banded fwd-bwd Gcells/s", "value": round(gcells, 4), "unit": "Gcells/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "reads_per_s": round(reads_per_s, 1),
        "end_to_end_reads_per_s": round(args.reads / e2e, 1),
        "config": {"workload": "BASELINE configs[2]: %d reads/GPU x (%d events x %d k-mers), "
                               "diagonalExpansion %d, 3-state strawMan signal HMM, per-read scaled "
                               "models, posterior decode" % (args.reads, args.events, args.kmers, args.band),
                   "cells_per_gpu": cells, "pairs_per_gpu": int(npairs.sum()),
                   "kernel": info,
                   "batches_in_flight": inflight,
                   "shader_clock_mhz_in_timed_region": round(float(np.mean(clock_mhz)), 0) if clock_mhz else None,
                   "pairs_finished_on_host_inside_timed_region": finalise,
                   "host_finalise_ms_per_step": round(1e3 * float(np.mean(fin_s)), 2) if fin_s else None,
                   "step_latency_ms": round(float(np.mean(kernel_ms)), 3),
                   "parallelism": "reads sharded over %d GPU(s), no collective" % world},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity_check": check,
        "host_prep_s": {"generate": round(t_gen, 2), "derive_models": round(t_models, 2),
                        "upload_and_band": round(t_upload, 2),
                        "end_to_end_note": "end_to_end_reads_per_s = reads / (derive_models + upload_and_band per "
                                           "batch + one step incl. the pairs on the host); synthetic generation "
                                           "is not part of it"},
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def kernel_stage(info, stage, cells):
    """forward-window / backward-window kernel figures from the per-window HIP events of the passes in `stage`"""
    if not stage:
        return None
    f_ms = float(np.mean([x[0] for x in stage]))
    k_ms = float(np.mean([x[1] for x in stage]))
    n_l = stage[0][2]
    if info.get("family") == "wave" and info.get("assembly_sweeps", 0) == 2:
        sfx = "_l%d" % info.get("cells_per_lane", 3)
        fwd, bwd = "cpecan_k_asm_forward" + sfx, "cpecan_k_asm_backward" + sfx
        extra = " (the window's cpecan_k_wv_post launch, totals and decode, runs on a stream of its own and is not in it)"
    elif info.get("family") == "wave":
        sfx = "_l%d" % info.get("cells_per_lane", 4)
        fwd, bwd = "cpecan_k_wv_forward" + sfx, "cpecan_k_wv_backward" + sfx
        extra = " (the backward figure includes the window's cpecan_k_wv_post launch: totals and decode)"
    else:
        sfx = {1: "_r1", 2: "_r2", 3: "_r3"}.get(info.get("waves_per_workgroup"), "")
        fwd, bwd = "cpecan_k_sy_forward" + sfx, "cpecan_k_sy_backward" + sfx
        extra = ""
    # the backward-window kernel re-reads the forward sweep's three values of every cell once (24 B per cell); the
    # forward-window kernel writes them once (24 B per cell).  The one with the longer launches is the dominant kernel.
    def fig(name, ms):
        return {"name": name, "launches_per_pass": n_l, "avg_launch_ms": round(ms / n_l, 4),
                "algorithmic_bytes_per_launch": round(cells * 24.0 / n_l),
                "achieved": round(cells * 24.0 / (ms / 1e3) / 1e9, 2),
                "frac": round(cells * 24.0 / (ms / 1e3) / 1e9 / 8000.0, 5)}
    f, k = fig(fwd, f_ms), fig(bwd, k_ms)
    note = ("launches of the batches in flight (and a batch's own forward and backward sweeps) overlap, so the sum of "
            "launch durations exceeds the pass time" + extra)
    return {"dominant_kernel": dict(f if f_ms > k_ms else k, note=note), "forward_kernel": f, "backward_kernel": k}


def bench_service(args, cp, bp, rank, local_rank, world, synth):
    """One-shot alignment of a stream of batches, end to end: every batch is new to the library -- its per-read model
    tables are derived (host libm) and uploaded, its band tables built, it is aligned once and its pairs are finished
    on the host.  A host thread prepares batch k + 1 while the GPU works on batch k.  Not a roofline figure: it measures
    what feeds the GPU.  --steps batches after --warmup; synthetic reads are generated up front."""
    import threading
    if world != 1:
        sys.exit("--mode service is a one-GPU measurement")
    n = args.warmup + args.steps
    data = [synth.make_batch(3 + 10 * k, args.reads, args.kmers, args.events, anchor_every=50) for k in range(min(n, 4))]
    family = "wave" if args.family == "auto" else args.family
    # (several one-shot batches alive at once: the wave family's assembly sweeps with their smaller footprint)
    flags = cp.FLAG_WORKGROUP_KERNELS if family == "workgroup" else cp.FLAG_SMALL_FOOTPRINT
    if family == "workgroup" and "CPECAN_SYSTOLIC_GROUPS" not in os.environ:
        os.environ["CPECAN_SYSTOLIC_GROUPS"] = "1"  # one stream group per batch: two batches' kernels overlap
    NSLOT = max(3, args.service_slots)  # one batch running, one queued behind it on the device, the others being prepared
    NPROD = max(1, args.service_threads)  # host threads that prepare and queue batches (the consumer is this thread)
    ctxs = [cp.Context(local_rank) for _ in range(NSLOT)]
    slot = [None] * NSLOT
    t_prep, t_models, t_batch = {}, {}, {}
    items = [make_items(cp, bt) for bt in data]
    queued = [threading.Event() for _ in range(n)]   # batch k has been handed to the device
    closed = [threading.Event() for _ in range(n)]   # batch k's results have been taken and its slot is free
    failed = []

    def prepare(k):
        t0 = time.perf_counter()
        bt, cx = data[k % len(data)], ctxs[k % NSLOT]
        cx.models_clear()
        if args.host_tables:  # the caller scaled every read's table itself (604 MB up per batch)
            cx.models_create([(cp.NANOPORE_TRANSITIONS, m, gx, gy) for (m, gx, gy) in bt["models"]])
        else:  # one pore model + five scaling parameters per read; rows assembled on the device
            cx.models_create_scaled((cp.NANOPORE_TRANSITIONS,) + bt["base_model"], bt["scalings"])
        t_models[k] = time.perf_counter() - t0
        t1 = time.perf_counter()
        slot[k % NSLOT] = cp.Batch(cx, items[k % len(data)], bt["x_chars"], bt["events"], bt["anchors"], bp,
                               cp.MODE_POSTERIOR, args.kernel, flags)
        t_batch[k] = time.perf_counter() - t1
        t_prep[k] = time.perf_counter() - t0

    def producer(first):
        # batches first, first + NPROD, ...: each is prepared as soon as its slot is free and queued in batch order
        try:
            for k in range(first, n, NPROD):
                if k >= NSLOT:
                    closed[k - NSLOT].wait()
                prepare(k)
                if k > 0:
                    queued[k - 1].wait()
                # wave kernels: one pass at a time, ordered on the device (no host round trip); workgroup kernels: the
                # queued batch's kernels overlap the running one's
                slot[k % NSLOT].run(after=slot[(k - 1) % NSLOT] if (k > 0 and family == "wave") else None)
                queued[k].set()
        except BaseException as e:  # noqa: BLE001 -- the consumer must not wait for ever
            failed.append(e)
            for ev in queued + closed:
                ev.set()

    threads = [threading.Thread(target=producer, args=(f,), daemon=True) for f in range(min(NPROD, n))]
    for th in threads:
        th.start()
    pairs = cells = 0
    t_start = None
    for k in range(n):
        if k == args.warmup:
            t_start = time.perf_counter()
            pairs = cells = 0
        # batch k is running or done and batch k + 1 queued behind it: the device goes from one pass to the next
        # without waiting for the host, whose producer threads meanwhile prepare the batches after it while this thread
        # (and the library's) fetches and finishes batch k's pairs
        queued[k].wait()
        if failed:
            raise failed[0]
        b = slot[k % NSLOT]
        b.sync()
        npairs, _, ncells = b.counts()
        pairs += int(npairs.sum())
        cells += int(ncells.sum())
        b.close()
        closed[k].set()
    for th in threads:
        th.join()
    t_prep, t_models, t_batch = ([d[k] for k in range(n)] for d in (t_prep, t_models, t_batch))
    elapsed = time.perf_counter() - t_start
    print(json.dumps({
        "metric": "end-to-end reads/s (one-shot alignment, host preparation overlapped)", "value": round(args.reads * args.steps / elapsed, 1),
        "unit": "reads/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "gcells_per_s": round(cells / elapsed / 1e9, 3),
        "config": {"workload": "a stream of BASELINE configs[2] batches (%d reads x %d events x %d k-mers), each prepared "
                               "from host buffers, aligned once, pairs finished on the host" % (args.reads, args.events, args.kmers),
                   "kernels": family + "-per-alignment", "slots": NSLOT, "producer_threads": NPROD,
                   "model_tables": "scaled by the caller, derived on the host" if args.host_tables else "scaled and assembled on the device (cpecan_hip_models_create_scaled)",
                   "host_prepare_ms_per_batch": round(1e3 * float(np.mean(t_prep[args.warmup:])), 1),
                   "of_which_model_tables_ms": round(1e3 * float(np.mean(t_models[args.warmup:])), 1),
                   "of_which_batch_create_ms": round(1e3 * float(np.mean(t_batch[args.warmup:])), 1),
                   "pairs": pairs}}), flush=True)
    return 0


def hdp_reads(n, lX, lY, seed, desc):
    """Synthetic reads for the HDP machine: random ACGT reference, per k-mer 0 events with p = 0.1, else 1 + stays so
    that about lY events come out; an event's mean is drawn around the mode of its k-mer's density in the HDP (the
    emission model the machine scores with); anchors every 50 k-mers on the true path."""
    import ctypes as C
    G, rows = desc.grid_length, int(desc.n_rows)
    grid = np.ctypeslib.as_array(C.cast(desc.grid, C.POINTER(C.c_double)), (G,))
    y = np.ctypeslib.as_array(C.cast(desc.posterior_predictive, C.POINTER(C.c_double)), (rows, G))
    A = desc.alphabet_size
    kmer_row = np.ctypeslib.as_array(C.cast(desc.kmer_row, C.POINTER(C.c_int32)), (A ** 6,))
    mode = grid[np.argmax(y, axis=1)]
    digit = {ch: desc.alphabet.decode().index(ch) for ch in "ACGT"}
    xs, evs, ans, items = [], [], [], []
    xo = yo = ao = 0
    for r in range(n):
        rng = np.random.default_rng(seed * 1000 + r)
        seq = rng.integers(0, 4, lX + 5)
        chars = np.frombuffer(b"ACGT", np.uint8)[seq]
        d = np.array([digit[c] for c in "ACGT"])[seq]
        kid = np.zeros(lX, np.int64)
        for j in range(6):
            kid = kid * A + d[j:j + lX]
        stay = max(0.05, 1.0 - 0.9 * lX / max(lY, 1))
        counts = np.where(rng.random(lX) < 0.10, 0, rng.geometric(1.0 - stay, lX))
        ev_k = np.repeat(np.arange(lX), counts)
        ev = np.zeros((ev_k.size, 3))
        ev[:, 0] = mode[kmer_row[kid[ev_k]]] + rng.normal(0, 1.0, ev_k.size)
        ev[:, 1] = np.abs(rng.normal(1.0, 0.2, ev_k.size)) + 1e-3
        ev[:, 2] = 0.01
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        emitting = np.flatnonzero(counts > 0)
        ax = emitting[np.searchsorted(emitting, np.arange(25, lX - 1, 50))]  # the next k-mer that emitted
        ax = np.unique(ax)
        an = np.stack([ax, first[ax]], axis=1).astype(np.int64)
        items.append(dict(x_offset=xo, lX=lX, y_offset=yo, lY=ev_k.size, anchor_offset=ao, n_anchors=len(an), model=0))
        xs.append(bytes(chars)); evs.append(ev); ans.append(an)
        xo += lX + 5; yo += ev_k.size; ao += len(an)
    return dict(x_chars=b"".join(xs), events=np.concatenate(evs), anchors=np.concatenate(ans), items=items)


def bench_hdp(args, cp, bp, rank, local_rank, world, dist, torch, sync_all):
    """BASELINE configs[4]: the HDP-emission signal machine on long reads (default 64 reads/GPU x 41 500 k-mers x
    ~50 000 events), read-sharded; the HDP comes from a serialized .nhdp through the host library's own reader."""
    import ctypes as C
    host = C.CDLL(os.path.join(ROOT, "cpecan-signal_amd", "libcpecan_host.so"))
    host.deserialize_nhdp.restype = C.c_void_p
    host.deserialize_nhdp.argtypes = [C.c_char_p]
    host.getHdpStateMachine3.restype = C.c_void_p
    host.getHdpStateMachine3.argtypes = [C.c_void_p]
    host.cpecan_hdp_machine_as_model.argtypes = [C.c_void_p, C.c_void_p]
    nh = host.deserialize_nhdp(args.hdp.encode())
    sm = host.getHdpStateMachine3(nh)
    desc = cp.HdpModelDesc()
    host.cpecan_hdp_machine_as_model(sm, C.byref(desc))
    # as in the default mode, consecutive steps work on distinct batches of --reads reads (--inflight 2): step s + 1
    # sweeps on the device (queued behind step s with cpecan_hip_batch_run_after: the wave kernels run one pass at a
    # time) while the host fetches and finishes the pairs of step s -- 9 GB of packed candidates per step cross PCIe
    nb = max(1, min(2, args.inflight, args.steps))
    t0 = time.time()
    bts = [hdp_reads(args.reads, args.kmers, args.events, 5 + 100 * rank + 7 * j, desc) for j in range(nb)]
    bt = bts[0]
    t_gen = time.time() - t0
    bs = []
    for j in range(nb):
        cx = cp.Context(local_rank)
        ids = np.zeros(1, np.int32)
        rc = cp.lib().cpecan_hip_modelsh_create(cx.h, C.byref(desc), 1, ids.ctypes.data_as(C.c_void_p))
        if rc != 0:
            sys.exit("cpecan_hip_modelsh_create: %s" % cp.lib().cpecan_hip_last_error().decode())
        bs.append(cp.Batch(cx, make_items(cp, bts[j]), bts[j]["x_chars"], bts[j]["events"], bts[j]["anchors"], bp, hdp=True))
    b = bs[0]
    ms = []

    def finish(j):
        bs[j].sync()
        ms.append(bs[j].elapsed_ms()[1])
        bs[j].counts()  # the pairs on the host as the reference's integers, inside the timed region

    def run_steps(n):
        pending = [False] * nb
        for s_ in range(n):
            j = s_ % nb
            if pending[j]:
                finish(j)
            bs[j].run(after=bs[(s_ - 1) % nb] if (nb > 1 and s_ > 0) else None)
            pending[j] = True
        for k in range(nb):  # the oldest first
            j = (n + k) % nb
            if pending[j]:
                finish(j)

    run_steps(max(args.warmup, nb))
    sync_all()
    ms = []
    t_start = time.perf_counter()
    run_steps(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t_start
    npairs = np.zeros(0, np.int64)
    cells_of = []
    for bb in bs:
        npj, _, ncj = bb.counts()
        cells_of.append(int(ncj.sum()))
        if bb is b:
            npairs = npj
    cells = int(round(sum(cells_of[s_ % nb] for s_ in range(args.steps)) / args.steps))  # per step
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ct = torch.tensor([cells], dtype=torch.int64, device="cuda")
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        total_cells = int(ct.item())
    else:
        total_cells = cells
    if rank == 0:
        achieved = cells * 48.0 / (elapsed / args.steps) / 1e9
        print(json.dumps({
            "metric": "banded fwd-bwd Gcells/s", "value": round(total_cells * args.steps / elapsed / 1e9, 4),
            "unit": "Gcells/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "reads_per_s": round(args.reads * world * args.steps / elapsed, 2),
            "config": {"workload": "BASELINE configs[4]: HDP-emission signal HMM (%s), %d reads/GPU x %d k-mers x ~%d "
                                   "events, diagonalExpansion %d, posterior decode"
                                   % (os.path.basename(args.hdp), args.reads, args.kmers,
                                      int(np.mean([it["lY"] for it in bt["items"]])), args.band),
                       "cells_per_gpu": cells, "pairs_per_gpu": int(npairs.sum()), "kernel": b.info(),
                       "batches_in_flight": nb,
                       "kernel_ms_per_step": round(float(np.mean(ms)), 3), "generate_s": round(t_gen, 2),
                       "parallelism": "reads sharded over %d GPU(s), no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": None,
                         "scope": "whole pass incl. the pairs on the host, 48 B per cell; kernels: %s"
                                  % ("cpecan_k_wv_*_h%d (wave per alignment)" % b.info().get("cells_per_lane", 0)
                                     if b.info()["kernel"] == "systolic" else "cpecan_k_generalh")},
            "cpu_baseline": None}), flush=True)
    for bb in bs:
        bb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def bench_em(args, cp, em, bp, rank, local_rank, world, dist, torch, synth, sync_all, first_batch, cpu_all):
    """BASELINE configs[3]: Baum-Welch iterations; per rank --reads reads (weak scaling), two concurrent batches"""
    bt = first_batch if first_batch is not None else \
        synth.make_batch(3 + 100 * rank, args.reads, args.kmers, args.events, anchor_every=50)
    t0 = time.time()
    ctxs = [cp.Context(local_rank) for _ in range(max(1, min(8, args.em_contexts)))]
    gap_x = np.full(4096, -2.3025850929940455)
    trans = np.array(cp.NANOPORE_TRANSITIONS, dtype=np.float64)
    e_step = em.PersistentEStep(cp, ctxs, bt, bp, range(len(bt["items"])), trans, gap_x,
                                dist if world > 1 else None, pseudocount=1e-4)
    t_setup = time.time() - t0
    state = {"t": trans, "g": gap_x, "lik": []}

    def iteration():
        e = e_step(state["t"], state["g"])      # E-step on the GPU + the all-reduce
        state["lik"].append(float(e[-1]))
        state["t"], state["g"] = em.m_step(e)   # normalise; the next E-step loads them in place on the device

    for _ in range(max(args.warmup, 1)):
        iteration()
    sync_all()
    state["lik"] = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        iteration()
    sync_all()
    elapsed = time.perf_counter() - t_start
    cells = int(sum(int(b.counts()[2].sum()) for _, b in e_step.batches))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ct = torch.tensor([cells], dtype=torch.int64, device="cuda")
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        total_cells = int(ct.item())
    else:
        total_cells = cells
    info = e_step.batches[0][1].info()
    e_step.close()
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        achieved = cells * 48.0 / (elapsed / args.steps) / 1e9
        print(json.dumps({
            "metric": "banded fwd-bwd Gcells/s", "value": round(total_cells * args.steps / elapsed / 1e9, 4),
            "unit": "Gcells/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "reads_per_s": round(args.reads * world * args.steps / elapsed, 1),
            "config": {"workload": "BASELINE configs[3]: Baum-Welch iterations, %d reads/GPU x (%d events x %d "
                                   "k-mers), diagonalExpansion %d; E-step on the GPU, one all-reduce of 4106 doubles "
                                   "per iteration, M-step in place on the device"
                                   % (args.reads, args.events, args.kmers, args.band),
                       "cells_per_gpu": cells, "kernel": info, "setup_s": round(t_setup, 2),
                       "running_likelihood": [round(v, 3) for v in state["lik"][:4]],
                       "parallelism": "reads sharded over %d GPU(s); all-reduce(SUM) of the expectation vector "
                                      "(RCCL)" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": None,
                         "scope": "whole iteration, 48 B per cell"},
            "cpu_baseline": cpu_all}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
