"""Baum-Welch loop for the strawMan signal machine on one or several GPUs (one process per GPU).

Mirrors the loop of the reference's trainer (scripts/trainModels.py:244-330; cPecanEm.py:107-209 for the
discrete machine): per iteration every worker computes expectations for its reads with the current model,
the expectations are summed, normalised (continuousPairHmm_normalize, impl/continuousHmm.c:174-191), loaded
back into the state machine (continuousPairHmm_loadTransitionsAndKmerGapProbs, :206-232) and the running
likelihood is logged.  The reference sums by writing one .expectations text file per worker and adding the
files up in Python (scripts/trainModels.py:126-135, scripts/nanoporeLib.py:991-1028); here each rank holds
[9 transitions | 4096 k-mer gap bins | likelihood] in HBM and ONE all-reduce(SUM) over RCCL combines them, after
which every rank normalises and reloads identically -- no file, no gather on a master.

This module is host-side control only: the E-step is the HIP path (binding.Batch in MODE_EXPECTATIONS); there
is no CPU fallback.  `e_step` may be replaced by a caller (the CPU tests substitute the oracle, tests/ only).
"""
import numpy as np

EXP_LEN = 9 + 4096 + 1


def shard(sizes, rank, world):
    """Indices of the reads rank `rank` works on: longest-first, dealt round-robin (LPT-like), so that the
    per-GPU cell counts differ by less than one read."""
    order = np.argsort(-np.asarray(sizes), kind="stable")
    return [int(i) for i in order[rank::world]]


def allreduce_expectations(vec, dist=None):
    """vec: torch tensor [..., EXP_LEN] float64 on the rank's device; summed in place over ranks."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec


def m_step(e):
    """continuousPairHmm_normalize (impl/continuousHmm.c:174-191) + continuousPairHmm_loadTransitionsAndKmerGapProbs
    (:206-232) on one expectation vector: the nine log transitions in the C-ABI's order and the 4096 log k-mer
    gap probabilities."""
    t = np.array(e[:9], dtype=np.float64).reshape(3, 3)
    t = t / t.sum(axis=1, keepdims=True)
    g = np.array(e[9:9 + 4096], dtype=np.float64)
    g = g / g.sum()
    with np.errstate(divide="ignore"):
        trans = np.array([
            np.log(t[0, 0]),        # MATCH_CONTINUE
            np.log(t[1, 0]),        # MATCH_FROM_GAP_X
            np.log(t[2, 0]),        # MATCH_FROM_GAP_Y
            np.log(t[0, 1]),        # GAP_OPEN_X
            np.log(t[0, 2]),        # GAP_OPEN_Y
            np.log(1 - t[1, 0]),    # GAP_EXTEND_X (sic: log(1 - P(gapX->match)), :217)
            np.log(t[2, 2]),        # GAP_EXTEND_Y
            np.log(t[2, 1]),        # GAP_SWITCH_TO_X
            -np.inf,                # GAP_SWITCH_TO_Y (:218)
        ])
        gap_x = np.log(g)
    return trans, gap_x


class _DeviceDoubles:
    """a device buffer of float64 as torch can adopt it (no copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2}


def gpu_e_step(cp, ctx, batch, bp, read_idx, transitions, gap_x, dist=None, pseudocount=0.0, ctx2=None):
    """Expectations of reads `read_idx` of `batch` (tests/synth.make_batch layout: per-read scaled match tables)
    under (transitions, gap_x): expectation batches in MODE_EXPECTATIONS, the per-model sums added on the device,
    one all-reduce over the ranks, one copy to the host.  Returns the EXP_LEN vector (identical on every rank).

    With a second context (`ctx2`, same device) the reads are dealt into two batches that run at the same time on
    the two contexts' streams, each as one stream group: the kernels of one batch fill the workgroup slots the
    other leaves idle (the same overlap bench.py uses between consecutive steps)."""
    import os
    import torch

    ctxs = [ctx] if ctx2 is None else [ctx, ctx2]
    dev = torch.device("cuda", ctx.device)
    # (once per read, as the reference's one process per read does: independent of how the reads are spread over ranks)
    total = torch.full((EXP_LEN,), float(pseudocount) * len(read_idx), dtype=torch.float64, device=dev)
    total[-1] = 0.0
    parts = [list(read_idx)[k::len(ctxs)] for k in range(len(ctxs))]
    saved = os.environ.get("CPECAN_SYSTOLIC_GROUPS")
    if len(ctxs) > 1 and saved is None:
        os.environ["CPECAN_SYSTOLIC_GROUPS"] = "1"
    running = []
    try:
        for cx, part in zip(ctxs, parts):
            if not part:
                continue
            items = np.zeros(len(part), cp.ITEM_DTYPE)
            models = []
            for k, i in enumerate(part):
                it = batch["items"][i]
                match, _, gap_y = batch["models"][it["model"]]
                models.append((list(transitions), match, gap_x, gap_y))
                items[k] = (it["x_offset"], it["lX"], it["y_offset"], it["lY"], it["anchor_offset"],
                            it["n_anchors"], k, 1, 1, 0)
            cx.models_clear()
            cx.models_create(models)
            b = cp.Batch(cx, items, batch["x_chars"], batch["events"], batch["anchors"], bp, cp.MODE_EXPECTATIONS,
                         cp.KERNEL_AUTO, cp.FLAG_WORKGROUP_KERNELS if len(ctxs) > 1 else 0)
            b.run()  # asynchronous on this context's streams: the next batch is issued before this one is waited for
            running.append(b)
        for b in running:
            b.sync()
            ptr, n = b.expectations_device_ptr()
            total += torch.as_tensor(_DeviceDoubles(ptr, n), device=dev).view(-1, EXP_LEN).sum(0)
        torch.cuda.synchronize(dev)
    finally:
        for b in running:
            b.close()
        if len(ctxs) > 1 and saved is None:
            os.environ.pop("CPECAN_SYSTOLIC_GROUPS", None)
    allreduce_expectations(total, dist)
    return total.cpu().numpy()


class PersistentEStep:
    """The E-step of a Baum-Welch run whose reads stay the same from iteration to iteration (the usual case): the
    inputs, band tables, rings and the per-read scaled emission tables are set up ONCE; an iteration rewrites the
    nine transitions and the 4096 k-mer gap probabilities in place on the device
    (cpecan_hip_models_set_transitions) and runs the batches again.  Calling the object is the `e_step` of train().

    contexts: one or two binding.Context on the same device; with two, the reads are dealt into two batches that
    run concurrently, each as one stream group."""

    def __init__(self, cp, contexts, batch, bp, read_idx, transitions, gap_x, dist=None, pseudocount=0.0):
        import os
        import torch
        self.cp, self.ctxs, self.dist, self.pseudocount = cp, list(contexts), dist, float(pseudocount)
        self.n_reads = len(list(read_idx))
        self.dev = torch.device("cuda", self.ctxs[0].device)
        saved = os.environ.get("CPECAN_SYSTOLIC_GROUPS")
        if len(self.ctxs) > 1 and saved is None:
            os.environ["CPECAN_SYSTOLIC_GROUPS"] = "1"
        self.batches = []
        try:
            for k, cx in enumerate(self.ctxs):
                part = list(read_idx)[k::len(self.ctxs)]
                if not part:
                    continue
                items = np.zeros(len(part), cp.ITEM_DTYPE)
                models = []
                for j, i in enumerate(part):
                    it = batch["items"][i]
                    match, _, gap_y = batch["models"][it["model"]]
                    models.append((list(transitions), match, gap_x, gap_y))
                    items[j] = (it["x_offset"], it["lX"], it["y_offset"], it["lY"], it["anchor_offset"],
                                it["n_anchors"], j, 1, 1, 0)
                cx.models_clear()
                cx.models_create(models)
                # two batches at once: the workgroup-per-alignment kernels (smaller footprint per CU) interleave;
                # one batch alone: the wave-per-alignment kernels overlap its own forward and backward sweeps
                self.batches.append((cx, cp.Batch(cx, items, batch["x_chars"], batch["events"], batch["anchors"], bp,
                                                  cp.MODE_EXPECTATIONS, cp.KERNEL_AUTO,
                                                  cp.FLAG_WORKGROUP_KERNELS if len(self.ctxs) > 1 else 0)))
        finally:
            if len(self.ctxs) > 1 and saved is None:
                os.environ.pop("CPECAN_SYSTOLIC_GROUPS", None)

    def __call__(self, transitions, gap_x):
        import torch
        total = torch.full((EXP_LEN,), self.pseudocount * self.n_reads, dtype=torch.float64, device=self.dev)
        total[-1] = 0.0
        for cx, b in self.batches:
            cx.models_set_transitions(transitions, gap_x)
            b.run()
        for _, b in self.batches:
            b.sync()
            ptr, n = b.expectations_device_ptr()
            total += torch.as_tensor(_DeviceDoubles(ptr, n), device=self.dev).view(-1, EXP_LEN).sum(0)
        torch.cuda.synchronize(self.dev)
        allreduce_expectations(total, self.dist)
        return total.cpu().numpy()

    def close(self):
        for _, b in self.batches:
            b.close()
        self.batches = []


def train(e_step, transitions, gap_x, iterations, log=None):
    """The loop: e_step(transitions, gap_x) -> summed EXP_LEN vector (already reduced over ranks).
    Returns dict(transitions, gap_x, running_likelihoods); logs 'i| likelihood' lines like the reference."""
    transitions = np.array(transitions, dtype=np.float64)
    gap_x = np.array(gap_x, dtype=np.float64)
    running = []
    for i in range(iterations):
        e = e_step(transitions, gap_x)
        running.append(float(e[-1]))
        transitions, gap_new = m_step(e)
        gap_x = gap_new
        if log is not None:
            log("%d| %f" % (i, running[-1]))
    return dict(transitions=transitions, gap_x=gap_x, running_likelihoods=running)
