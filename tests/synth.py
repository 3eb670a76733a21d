"""Seeded synthetic nanopore workloads of the shapes named in BASELINE.json.

Used by the parity tests and by bench.py.  Pure numpy; no reference data files are needed.
A "read" here is one strand: lX reference k-mers (lX+5 nucleotides), lY events that walk the k-mers
with skips and stays, a per-read affine rescaling of the pore model, and guide anchors on the true
path.
"""
import numpy as np

NUM_KMERS = 4096
MODEL_LEN = 1 + NUM_KMERS * 5
SEED0 = 0xC0FFEE


def synthetic_pore_model(seed=SEED0):
    """(match[20481], gap_x[4096], gap_y[20481]) in the reference's .model layout
    (level_mean, level_sd, noise_mean, noise_sd, noise_lambda per k-mer; element 0 = correlation)."""
    rng = np.random.default_rng(seed)
    match = np.zeros(MODEL_LEN)
    t = match[1:].reshape(NUM_KMERS, 5)
    t[:, 0] = rng.uniform(45.0, 75.0, NUM_KMERS)
    t[:, 1] = rng.uniform(0.6, 1.6, NUM_KMERS)
    t[:, 2] = rng.uniform(0.5, 1.5, NUM_KMERS)
    t[:, 4] = rng.uniform(3.0, 12.0, NUM_KMERS)
    t[:, 3] = np.sqrt(t[:, 2] ** 3 / t[:, 4])
    gap_y = match.copy()
    gap_y[1:].reshape(NUM_KMERS, 5)[:, 1] *= 1.75  # ratio of lines 1 and 3 of template_median68pA.model
    gap_x = np.full(NUM_KMERS, -2.3025850929940455)  # log(0.1), stateMachine.c:1506
    return match, gap_x, gap_y


def scale_model(match, scale, shift, var, scale_sd, var_sd):
    """emissions_signal_scaleModel (stateMachine.c:631-651) on a copy of the match table."""
    m = match.copy()
    t = m[1:].reshape(NUM_KMERS, 5)
    t[:, 0] = t[:, 0] * scale + shift
    t[:, 1] = t[:, 1] * var
    t[:, 2] = t[:, 2] * scale_sd
    t[:, 4] = t[:, 4] * var_sd
    t[:, 3] = np.sqrt(np.power(t[:, 2], 3.0) / t[:, 4])
    return m


def kmer_indices(seq_bytes):
    lut = np.full(256, -1, np.int64)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    b = lut[np.frombuffer(seq_bytes, np.uint8)]
    n = len(seq_bytes) - 5
    idx = np.zeros(n, np.int64)
    for j in range(6):
        idx = idx * 4 + b[j:j + n]
    return idx


def make_read(rng, match, lX, lY, anchor_every=50, jitter=2):
    """One strand.  Returns dict(seq, events[lY,3], anchors[n,2], scale_params, true_y)."""
    seq = rng.integers(0, 4, lX + 5).astype(np.uint8)
    seq_bytes = bytes(np.frombuffer(b"ACGT", np.uint8)[seq])
    kidx = kmer_indices(seq_bytes)
    # events per k-mer: 0 with p=.1 (skip), else 1 + Geometric stays; then forced to sum to lY
    stay = max(0.05, min(0.9, 1.0 - 0.9 * lX / max(lY, 1))) if lY > 0.9 * lX else 0.05
    counts = np.where(rng.random(lX) < 0.10, 0, rng.geometric(1.0 - stay, lX))
    diff = int(lY - counts.sum())
    while diff != 0:
        if diff > 0:
            np.add.at(counts, rng.integers(0, lX, diff), 1)
        else:
            nz = np.flatnonzero(counts > 0)
            pick = rng.choice(nz, size=min(-diff, nz.size), replace=False)
            counts[pick] -= 1
        diff = int(lY - counts.sum())
    ev_kmer = np.repeat(np.arange(lX), counts)  # k-mer position of every event
    scale, shift = rng.uniform(0.95, 1.05), rng.uniform(-5.0, 5.0)
    var, scale_sd, var_sd = rng.uniform(0.9, 1.1), rng.uniform(0.9, 1.2), rng.uniform(0.9, 1.2)
    scaled = scale_model(match, scale, shift, var, scale_sd, var_sd)
    t = scaled[1:].reshape(NUM_KMERS, 5)[kidx[ev_kmer]]
    events = np.zeros((lY, 3))
    events[:, 0] = rng.normal(t[:, 0], t[:, 1])
    events[:, 1] = np.maximum(np.abs(rng.normal(t[:, 2], t[:, 3])), 1e-3)
    events[:, 2] = rng.exponential(0.01, lY)
    # anchors: every anchor_every-th k-mer that emitted, at its first event, jittered, increasing
    first_ev = np.concatenate([[0], np.cumsum(counts)[:-1]])
    anchors = []
    px, py = -1, -1
    for x0 in range(anchor_every // 2, lX, anchor_every):
        x = x0
        while x < lX and counts[x] == 0:  # the anchor sits on the true path: next k-mer that emitted
            x += 1
        if x >= lX:
            break
        y = int(first_ev[x]) + int(rng.integers(-jitter, jitter + 1))
        y = min(max(y, 0), lY - 1)
        if x > px and y > py:
            anchors.append((x, y))
            px, py = x, y
    return dict(seq=seq_bytes, events=events, anchors=np.array(anchors, np.int64).reshape(-1, 2),
                scaled_match=scaled, scale_params=(scale, shift, var, scale_sd, var_sd),
                ev_kmer=ev_kmer)


def make_batch(config_id, n_reads, lX, lY, anchor_every=50, model_seed=SEED0, distinct_models=True,
               length_sigma=0.0):
    """Concatenated inputs for the C-ABI batch entry point.

    Returns dict with x_chars (bytes), events [N,3], anchors [A,2], items (list of dicts with
    x_offset,lX,y_offset,lY,anchor_offset,n_anchors,model) and models (list of
    (match, gap_x, gap_y)) -- one scaled model per read when distinct_models, else one shared; base_model (the
    unscaled pore model) and scalings [n_reads, 5] (every read's scale, shift, var, scale_sd, var_sd).
    """
    match, gap_x, gap_y = synthetic_pore_model(model_seed)
    xs, evs, ans, items, models, scalings = [], [], [], [], [], []
    xo = yo = ao = 0
    for r in range(n_reads):
        rng = np.random.default_rng(SEED0 + config_id * 1000 + r)
        lx, ly = lX, lY
        if length_sigma > 0:
            f = float(np.clip(rng.lognormal(0.0, length_sigma), 0.125, 3.75))
            ly = max(64, int(lY * f))
            lx = max(32, int(lX * f))
        rd = make_read(rng, match, lx, ly, anchor_every)
        xs.append(rd["seq"])
        evs.append(rd["events"])
        ans.append(rd["anchors"])
        scalings.append(rd["scale_params"])
        if distinct_models:
            models.append((rd["scaled_match"], gap_x, gap_y))
        items.append(dict(x_offset=xo, lX=lx, y_offset=yo, lY=ly, anchor_offset=ao,
                          n_anchors=len(rd["anchors"]), model=r if distinct_models else 0))
        xo += lx + 5
        yo += ly
        ao += len(rd["anchors"])
    if not distinct_models:
        models.append((match, gap_x, gap_y))
    return dict(x_chars=b"".join(xs), events=np.concatenate(evs), anchors=np.concatenate(ans),
                items=items, models=models, base_model=(match, gap_x, gap_y), scalings=np.array(scalings))
