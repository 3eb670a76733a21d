"""Read sharding and the Baum-Welch reduce step for N ranks (one process per GPU).

The reference reduces sufficient statistics by writing one .expectations text file per worker and summing
them in Python (scripts/trainModels.py:126-135, cPecanEm.py:37-51).  Here every rank holds a vector
[9 transitions | 4096 k-mer gap bins | likelihood] per model and ONE all-reduce(SUM) combines them
(RCCL over xGMI on the GPU node, gloo in the CPU tests); every rank then normalises and reloads the
model identically.  Aligned-pair decode needs no collective: each rank keeps its own reads.
"""
import numpy as np

EXP_LEN = 9 + 4096 + 1


def shard(sizes, rank, world):
    """Indices of the reads rank `rank` works on: longest-first, dealt round-robin (LPT-like)."""
    order = np.argsort(-np.asarray(sizes), kind="stable")
    return [int(i) for i in order[rank::world]]


def allreduce_expectations(vec, dist=None):
    """vec: torch tensor [n_models, EXP_LEN] float64 on the rank's device; summed in place over ranks."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec


def m_step(e):
    """continuousPairHmm_normalize (impl/continuousHmm.c:174-191) + the transition part of
    continuousPairHmm_loadTransitionsAndKmerGapProbs (:206-232) on one expectation vector."""
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
