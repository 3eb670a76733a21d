/*
 * cpecan_hip.h -- C-ABI of the MI355X (gfx950) implementation of cPecan's banded pair-HMM
 * forward / backward / posterior DP over nanopore events x reference k-mers.
 *
 * Plain C: opaque handles, plain pointers and sizes, no HIP or C++ types in any signature.
 * This is the boundary the reference's C code would link against for this path; each entry point
 * names the reference interface it replaces.  All functions return CPECAN_OK (0) or a negative
 * CPECAN_E* code; cpecan_hip_last_error() gives the message of the calling thread's last failure.
 * There is no CPU fallback: without a usable GPU every compute entry point fails with
 * CPECAN_ENODEVICE.
 */
#ifndef CPECAN_HIP_H_
#define CPECAN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPECAN_OK 0
#define CPECAN_ENODEVICE (-1)  /* no HIP device / runtime error at start-up          */
#define CPECAN_EINVAL (-2)     /* bad argument                                       */
#define CPECAN_EHIP (-3)       /* a HIP call failed (message in last_error)          */
#define CPECAN_EOVERFLOW (-4)  /* an output capacity was too small                   */
#define CPECAN_EBAND (-5)      /* anchors describe an invalid band (diagonal_construct throw) */

#define CPECAN_NUM_KMERS 4096
#define CPECAN_KMER_LENGTH 6
#define CPECAN_MODEL_PARAMS 5
#define CPECAN_MODEL_TABLE_LEN (1 + CPECAN_NUM_KMERS * CPECAN_MODEL_PARAMS)
#define CPECAN_PAIR_ALIGNMENT_PROB_1 10000000 /* inc/pairwiseAligner.h:26 */

typedef struct cpecan_ctx cpecan_ctx;     /* one per (host thread, GPU): device, stream, models */
typedef struct cpecan_batch cpecan_batch; /* a set of independent alignments resident in HBM    */

/* ---- start-up ------------------------------------------------------------------------------ */
int cpecan_hip_device_count(int *count);
int cpecan_hip_ctx_create(int device, cpecan_ctx **ctx);
int cpecan_hip_ctx_destroy(cpecan_ctx *ctx);
const char *cpecan_hip_last_error(void);
const char *cpecan_hip_version(void);

/* ---- model ---------------------------------------------------------------------------------
 * Replaces the StateMachine3 the reference builds with getStrawManStateMachine3()
 * (impl/stateMachine.c:1725) and rescales per read with emissions_signal_scaleModel() (:631):
 * transitions in the order of struct _StateMachine3 (inc/stateMachine.h:179-187), emission tables
 * in the reference's own layout (EMISSION_MATCH_PROBS / EMISSION_GAP_Y_PROBS = 1+4096*5 doubles,
 * EMISSION_GAP_X_PROBS = 4096 doubles).  The tables are read once, on the host, to derive the
 * device table (log sigma etc. use the host libm, as the reference's per-cell log() does). */
typedef struct {
    double transitions[9]; /* MATCH_CONTINUE, MATCH_FROM_GAP_X, MATCH_FROM_GAP_Y, GAP_OPEN_X,
                              GAP_OPEN_Y, GAP_EXTEND_X, GAP_EXTEND_Y, GAP_SWITCH_TO_X, GAP_SWITCH_TO_Y */
    const double *match_probs; /* [CPECAN_MODEL_TABLE_LEN] */
    const double *gap_x_probs; /* [CPECAN_NUM_KMERS]       */
    const double *gap_y_probs; /* [CPECAN_MODEL_TABLE_LEN] */
} cpecan_sm3_model;

/* Derives and uploads n models (host work is spread over `threads` OS threads, <=0: all cores);
 * ids[i] receives the handle of models[i]. */
int cpecan_hip_models_create(cpecan_ctx *ctx, const cpecan_sm3_model *models, int32_t n,
                             int32_t threads, int32_t *ids);
/* The same for n reads that share one pore model and differ by their scaling parameters only -- what
 * emissions_signal_scaleModel (impl/stateMachine.c:631-651) does to a read's copy of EMISSION_MATCH_PROBS before the
 * alignment (vanillaAlign.c:624-640): level_mean * scale + shift, level_sd * var, noise_mean * scale_sd,
 * noise_lambda * var_sd, noise_sd = sqrt(noise_mean^3 / noise_lambda).  The host takes only what needs its libm
 * (pow, sqrt and the two logs per k-mer) and uploads three doubles per k-mer per read; the device assembles the rows
 * with single IEEE operations.  The resulting device tables are bit-identical to cpecan_hip_models_create() on the
 * n host-scaled tables (tests/test_scaled_models_gpu.py), at a sixth of the upload. */
typedef struct {
    double scale, shift, var, scale_sd, var_sd;
} cpecan_read_scaling;
int cpecan_hip_models_create_scaled(cpecan_ctx *ctx, const cpecan_sm3_model *base, const cpecan_read_scaling *scalings,
                                    int32_t n, int32_t threads, int32_t *ids);
/* Test aid: copies the derived device table of strawMan model `id` to out (n_doubles receives its length; out may be
 * NULL to query the length only). */
int cpecan_hip_models_download(cpecan_ctx *ctx, int32_t id, double *out, int64_t capacity, int64_t *n_doubles);
int cpecan_hip_models_clear(cpecan_ctx *ctx);
/* The M-step of Baum-Welch changes the nine transitions and the k-mer gap probabilities only
 * (continuousPairHmm_loadTransitionsAndKmerGapProbs impl/continuousHmm.c:206-232); the per-read scaled emission
 * tables stay.  Rewrites those 9 + 4096 values in every strawMan model of the context in place (gap_x_probs may be
 * NULL: transitions only), so that batches created on the context can simply be run again for the next
 * iteration -- no table derivation, no upload. */
int cpecan_hip_models_set_transitions(cpecan_ctx *ctx, const double *transitions /*[9]*/,
                                      const double *gap_x_probs /*[CPECAN_NUM_KMERS] or NULL*/);

/* The 5-state symbol machine of DNA-against-DNA alignment: stateMachine5_construct(fiveState)
 * (impl/stateMachine.c:896-965; BASELINE configs[0]).  transitions in the order of struct
 * _StateMachine5 (inc/stateMachine.h:108-124): MATCH_CONTINUE, MATCH_FROM_SHORT_GAP_X,
 * MATCH_FROM_LONG_GAP_X, GAP_SHORT_OPEN_X, GAP_SHORT_EXTEND_X, GAP_SHORT_SWITCH_TO_X, GAP_LONG_OPEN_X,
 * GAP_LONG_EXTEND_X, GAP_LONG_SWITCH_TO_X, then the same eight for Y; emissions as
 * emissions_symbol_* reads them (:155-173): match [4x4] by (x base, y base), gaps [4].
 * Ids live in their own space (used by cpecan_hip_batch_create_dna only). */
typedef struct {
    double transitions[17];
    double match_probs[16];
    double gap_x_probs[4];
    double gap_y_probs[4];
} cpecan_sm5_model;
int cpecan_hip_models5_create(cpecan_ctx *ctx, const cpecan_sm5_model *models, int32_t n, int32_t *ids);

/* The 4-state signal machine: getStateMachine4() (impl/stateMachine.c:1750-1759, stateMachine4_cellCalculate :867-897)
 * after emissions_signal_scaleModel(): match, short gap X, short gap Y, long gap X over the strawMan emissions.
 * transitions in the member order of struct _StateMachine4 (inc/stateMachine.h:134-152): MATCH_CONTINUE,
 * MATCH_FROM_SHORT_GAP_X, MATCH_FROM_LONG_GAP_X, MATCH_FROM_SHORT_GAP_Y, GAP_SHORT_OPEN_X, GAP_SHORT_EXTEND_X,
 * GAP_SHORT_OPEN_Y, GAP_SHORT_EXTEND_Y, GAP_LONG_OPEN_X, GAP_LONG_EXTEND_X, GAP_LONG_SWITCH_TO_X; tables as in the
 * strawMan model above (the machine's k-mer gap table is all zeros as getStateMachine4 leaves it). */
typedef struct {
    double transitions[11];
    const double *match_probs; /* [CPECAN_MODEL_TABLE_LEN] */
    const double *gap_x_probs; /* [CPECAN_NUM_KMERS]       */
    const double *gap_y_probs; /* [CPECAN_MODEL_TABLE_LEN] */
} cpecan_sm4_model;
int cpecan_hip_models4_create(cpecan_ctx *ctx, const cpecan_sm4_model *models, int32_t n, int32_t *ids);

/* The 3-state "vanilla" signal machine: getSignalStateMachine3Vanilla() (impl/stateMachine.c:1761) after
 * emissions_signal_scaleModel().  Fields as in struct _StateMachine3Vanilla (inc/stateMachine.h:189-205):
 * the two transition fudge factors (stateMachine3Vanilla_setStrandTransitionsToDefaults :1291), the three
 * end-state log-probabilities, the match and extra-event tables in the reference's layout and the 60
 * skip-bin values of EMISSION_GAP_X_PROBS (beta[30] | alpha[30], :284-297).  X elements are read as
 * sequence_getKmer2 does (:320-325); k-mers must be ACGT-only (the reference computes NaN otherwise).
 * Ids live in their own space (used by cpecan_hip_batch_create_vanilla only). */
typedef struct {
    double m_to_y_not_x, e_to_e;
    double end_match_prob, end_from_x_prob, end_from_y_prob;
    const double *match_probs; /* [CPECAN_MODEL_TABLE_LEN] */
    const double *skip_probs;  /* [60]                     */
    const double *gap_y_probs; /* [CPECAN_MODEL_TABLE_LEN] */
} cpecan_vanilla_model;
int cpecan_hip_modelsv_create(cpecan_ctx *ctx, const cpecan_vanilla_model *models, int32_t n,
                              int32_t threads, int32_t *ids);

/* The 3-state HDP signal machine: getHdpStateMachine3(NanoporeHDP *) (impl/stateMachine.c:1738) over a
 * finalized NanoporeHDP (deserialize_nhdp impl/nanopore_hdp.c:845).  transitions as cpecan_sm3_model; the
 * HDP as densities need it (dir_proc_density impl/hdp.c:2577-2601): the alphabet (sorted, as the k-mer ids
 * are taken over it, impl/nanopore_hdp.c:348-380), the sampling grid, and per k-mer id the row -- in
 * posterior_predictive / spline_slopes, rows x grid_length doubles -- of the Dirichlet process's nearest
 * OBSERVED ancestor.  Ids live in their own space (used by cpecan_hip_batch_create_hdp only); all models of
 * a context must share one alphabet. */
typedef struct {
    double transitions[9];
    const char *alphabet;
    int32_t alphabet_size;      /* <= 16 */
    int32_t grid_length;
    const double *grid;         /* [grid_length] */
    int64_t n_rows;
    const double *posterior_predictive; /* [n_rows * grid_length] */
    const double *spline_slopes;        /* [n_rows * grid_length] */
    const int32_t *kmer_row;    /* [alphabet_size ^ 6] */
} cpecan_hdp_model;
int cpecan_hip_modelsh_create(cpecan_ctx *ctx, const cpecan_hdp_model *models, int32_t n, int32_t *ids);

/* ---- band / split geometry (host integer code, exported because the reference exports it) ----
 * cpecan_band_construct: band_construct (impl/pairwiseAligner.c:132); xmyL/xmyR hold lX+lY+1 entries.
 * cpecan_split_points: getSplitPoints (:1313); out holds up to cap 4-tuples; returns the count. */
int cpecan_band_construct(const int64_t *anchors, int64_t n_anchors, int64_t lX, int64_t lY,
                          int64_t expansion, int32_t *xmyL, int32_t *xmyR);
int64_t cpecan_split_points(const int64_t *anchors, int64_t n_anchors, int64_t lX, int64_t lY,
                            int64_t max_matrix_size, int ragged_left, int ragged_right,
                            int64_t *out, int64_t cap);

/* ---- batch ---------------------------------------------------------------------------------
 * One work item = one getPosteriorProbsWithBanding() call (impl/pairwiseAligner.c:870): a k-mer
 * sequence X of lX elements (lX+5 nucleotides starting at x_chars[x_offset]), an event sequence Y
 * of lY elements (events[3*(y_offset+i)] = mean, noise, duration -- the reference's layout),
 * anchors relative to the item, ragged-end flags and a model.  Sub-alignments produced by
 * getSplitPoints are separate items that point into the same buffers. */
typedef struct {
    int64_t x_offset, lX;
    int64_t y_offset, lY;
    int64_t anchor_offset, n_anchors; /* into anchors[] as (x,y) pairs */
    int32_t model_id;
    int32_t ragged_left, ragged_right;
    int32_t reserved;
} cpecan_item;

/* the PairwiseAlignmentParameters fields this path reads (inc/pairwiseAligner.h:80-91) */
typedef struct {
    double threshold;
    int64_t minDiagsBetweenTraceBack;
    int64_t traceBackDiagonals;
    int64_t diagonalExpansion;
} cpecan_band_params;

#define CPECAN_MODE_POSTERIOR 0    /* diagonalCalculationPosteriorMatchProbs (:756) */
#define CPECAN_MODE_EXPECTATIONS 1 /* diagonalCalculation_Expectations (:841)       */

#define CPECAN_KERNEL_AUTO 0
#define CPECAN_KERNEL_GENERAL 1  /* any band width; diagonals live in HBM            */
#define CPECAN_KERNEL_SYSTOLIC 2 /* band <= 248 k-mers wide; register-resident wavefront */

#define CPECAN_FLAG_DEBUG_DUMP 1 /* keep forward/backward cells for cpecan_hip_batch_debug_cells */
#define CPECAN_FLAG_UNBANDED 2   /* getAlignedPairsWithoutBanding (:1512): full matrix, one traceback from
                                    the last diagonal, one totalProbability taken there; anchors and
                                    diagonalExpansion are ignored (general kernel only) */
#define CPECAN_FLAG_SCAN_DECODE 4 /* systolic kernel, diagnostic: decode posteriors by scanning every cell of
                                    the band instead of the sweep's candidate lists (the path a window falls
                                    back to by itself when its totals drift or a list overflows) */

#define CPECAN_FLAG_EXPECTATIONS 8 /* cpecan_hip_batch_create_dna / _vanilla / _hdp: run diagonalCalculation_Expectations
                                      (:841) instead of the posterior decode, as CPECAN_MODE_EXPECTATIONS does
                                      for the signal batches */

#define CPECAN_FLAG_WORKGROUP_KERNELS 16 /* register-resident path: run this batch on the workgroup-per-alignment
                                           kernels (cpecan_k_sy_*: 1..4 waves share an alignment through LDS) instead
                                           of the wave-per-alignment ones (cpecan_k_wv_*, the default).  Same results
                                           bit for bit; the workgroup family is the one that runs several batches at
                                           once well (more, smaller-footprint workgroups per CU).  The environment
                                           variable CPECAN_KERNELS=systolic asks the same for every batch. */

#define CPECAN_FLAG_SMALL_FOOTPRINT 64 /* a batch that may run the assembly sweeps keeps the two-window ring and the single
                                        * scratch of the compiled wave kernels (24 instead of 45 GB for 1024 reads of
                                        * 10k events x 5k k-mers): its post kernel then runs between its backward sweeps
                                        * (-10 % when batches are chained, nothing for a batch alone).  For callers that
                                        * keep several one-shot batches alive at once. */
#define CPECAN_FLAG_GENERAL_KERNEL 32 /* cpecan_hip_batch_create_hdp / _vanilla: keep the batch on the general kernel
                                         (any band width) instead of the wave-per-alignment kernels of that machine
                                         (posterior decode; bands <= 248 k-mers for the HDP machine, <= 184 for the
                                         vanilla one); same results bit for bit */

/* Copies the inputs to HBM and builds per-item band tables.  All host pointers may be released
 * after the call returns. */
int cpecan_hip_batch_create(cpecan_ctx *ctx, const cpecan_item *items, int64_t n_items,
                            const char *x_chars, int64_t n_x_chars, const double *events,
                            int64_t n_events, const int64_t *anchors, int64_t n_anchor_pairs,
                            const cpecan_band_params *params, int32_t mode, int32_t kernel,
                            int32_t flags, cpecan_batch **batch);
/* Runs the DP for every item (asynchronous on the context's stream). */
/* DNA against DNA with a 5-state model (getAlignedPairsUsingAnchors with a stateMachine5 and
 * sequence_getBase on both sides, impl/pairwiseAligner.c:1456,:308): X and Y are nucleotide strings,
 * lX / lY count bases, x_offset / y_offset index x_chars / y_chars, model_id is a
 * cpecan_hip_models5_create id.  General kernel; flags: DEBUG_DUMP is not available, UNBANDED and
 * EXPECTATIONS are (the latter: hmmDiscrete sums through cpecan_hip_batch_fetch_expectations). */
int cpecan_hip_batch_create_dna(cpecan_ctx *ctx, const cpecan_item *items, int64_t n_items,
                                const char *x_chars, int64_t n_x, const char *y_chars, int64_t n_y,
                                const int64_t *anchors, int64_t n_anchor_pairs,
                                const cpecan_band_params *params, int32_t flags, cpecan_batch **out);

/* k-mers against events with a vanilla model (getAlignedPairsUsingAnchors with a StateMachine3Vanilla,
 * sequence_getKmer2 / sequence_getEvent): same buffers as cpecan_hip_batch_create, model_id is a
 * cpecan_hip_modelsv_create id.  General kernel; flags: UNBANDED or EXPECTATIONS. */
int cpecan_hip_batch_create_vanilla(cpecan_ctx *ctx, const cpecan_item *items, int64_t n_items,
                                    const char *x_chars, int64_t n_x, const double *events, int64_t n_events,
                                    const int64_t *anchors, int64_t n_anchor_pairs,
                                    const cpecan_band_params *params, int32_t flags, cpecan_batch **out);

/* k-mers against events with a 4-state model (getAlignedPairsUsingAnchors / getAlignedPairsWithoutBanding with the
 * StateMachine of getStateMachine4, sequence_getKmer / sequence_getEvent): same buffers as cpecan_hip_batch_create,
 * model_id is a cpecan_hip_models4_create id.  General kernel, posterior decode; flags: UNBANDED. */
int cpecan_hip_batch_create_sm4(cpecan_ctx *ctx, const cpecan_item *items, int64_t n_items,
                                const char *x_chars, int64_t n_x, const double *events, int64_t n_events,
                                const int64_t *anchors, int64_t n_anchor_pairs,
                                const cpecan_band_params *params, int32_t flags, cpecan_batch **out);

/* k-mers against events with an HDP model (getAlignedPairsUsingAnchors with a StateMachine3_HDP,
 * sequence_getKmer3 / sequence_getEvent): same buffers as cpecan_hip_batch_create (x characters over the
 * model's alphabet), model_id is a cpecan_hip_modelsh_create id.  General kernel; flags: UNBANDED or
 * EXPECTATIONS. */
int cpecan_hip_batch_create_hdp(cpecan_ctx *ctx, const cpecan_item *items, int64_t n_items,
                                const char *x_chars, int64_t n_x, const double *events, int64_t n_events,
                                const int64_t *anchors, int64_t n_anchor_pairs,
                                const cpecan_band_params *params, int32_t flags, cpecan_batch **out);

int cpecan_hip_batch_run(cpecan_batch *batch);
/* The same, but the batch's kernels start when the last run of `after` (a batch of another context on the same device;
 * NULL: no condition) has finished -- ordered on the device, no host round trip.  A stream of batches then keeps the
 * device busy with one pass at a time (the wave-per-alignment kernels fill the register files with one batch) while
 * the host fetches and finishes the previous batch's pairs and prepares the next one. */
int cpecan_hip_batch_run_after(cpecan_batch *batch, cpecan_batch *after);
int cpecan_hip_batch_sync(cpecan_batch *batch);
/* HIP-event time of the last run's kernels, in ms (after sync). */
int cpecan_hip_batch_elapsed_ms(cpecan_batch *batch, float *ms_total, float *ms_dp_kernel);
/* The shader clock (MHz) the chip ran at during the forward sweeps of the batch's last run, from the sweeps' own cycle
 * and 100 MHz reference counters (wave-per-alignment kernels; 0 on the other kernels).  The sweeps are bound by
 * instruction issue, so their time follows this clock, and it differs between machines under this fp64 load. */
int cpecan_hip_batch_shader_clock_mhz(cpecan_batch *batch, double *mhz);
/* Which kernel the batch uses (CPECAN_KERNEL_GENERAL / _SYSTOLIC after AUTO is resolved), how many
 * workgroups it launches and the widest band (cells) among its items. */
int cpecan_hip_batch_info(cpecan_batch *batch, int32_t *kernel, int32_t *workgroups, int32_t *max_width);
/* Systolic path only: waves per workgroup of the kernel build the batch runs on -- the fewest whose 64 slots each
 * hold the widest band of the batch: 1 (bands up to 56 k-mers), 2 (120), 3 (184) or 4 (248).  The fewer waves an
 * alignment takes, the more alignments a CU holds (16, 8, 5, 4). */
int cpecan_hip_batch_systolic_rows(cpecan_batch *batch, int32_t *rows);
/* Register-resident path only: *wave = 1 if the batch runs on the wave-per-alignment kernels (rows is then the
 * number of cells a lane holds: 2, 3 or 4), 0 on the workgroup-per-alignment ones (rows = waves per workgroup).  For a
 * DNA batch: 1 if its posterior decode runs on the one-wave-per-alignment 5-state kernel (bands up to 192 cells), 0 if
 * on the general one. */
int cpecan_hip_batch_kernel_family(cpecan_batch *batch, int32_t *wave);
/* *sweeps = 1 if the batch's sweeps run on the hand-scheduled assembly kernels (strawMan machine, posterior decode,
 * bands of 121..158 k-mers: the BASELINE configs[2] shape), 2 if both the forward and the backward sweep do, 0 if on
 * the compiled kernels.  CPECAN_ASM=0 in the environment keeps every batch on the compiled kernels. */
int cpecan_hip_batch_assembly_sweeps(cpecan_batch *batch, int32_t *sweeps);
/* Systolic path only: HIP-event time of the last run spent in the forward-window kernels and in
 * the backward-window kernels (each launched `launches_each` times, once per traceback window). */
int cpecan_hip_batch_stage_ms(cpecan_batch *batch, float *ms_forward, float *ms_backward,
                              int32_t *launches_each);
/* Per-item result sizes: aligned pairs, refreshes of totalProbability, in-band cells. */
int cpecan_hip_batch_counts(cpecan_batch *batch, int64_t *n_pairs, int64_t *n_totals,
                            int64_t *n_cells);
/* Aligned pairs of one item as (floor(p*1e7), x, y) int64 triples in the order the reference's
 * diagonalPosteriorProbFn emits them (windows forward, diagonals descending, x-y ascending);
 * logp (may be NULL) receives (F+B)-total for each. */
int cpecan_hip_batch_fetch_pairs(cpecan_batch *batch, int64_t item, int64_t *triples, double *logp,
                                 int64_t cap);
/* totalProbability refreshes of one item: the diagonal and the value, in the order computed. */
int cpecan_hip_batch_fetch_totals(cpecan_batch *batch, int64_t item, int64_t *xay, double *total,
                                  int64_t cap);
/* Expectations (mode EXPECTATIONS): per model id, 9 transitions [from*3+to] + 4096 k-mer gap bins
 * + likelihood = 4106 doubles, summed over the items of the batch that use that model. The device
 * buffer pointer is exposed so that a caller can all-reduce it in place (RCCL) before fetching. */
#define CPECAN_EXPECTATION_LEN (9 + CPECAN_NUM_KMERS + 1)
/* DNA batches (HmmDiscrete, impl/discreteHmm.c:10-153): 25 transitions [from*5+to], 5 x 16 emissions
 * [state*16 + x*4 + y] (cell_updateExpectations :407-424) + likelihood, per cpecan_hip_models5_create id */
#define CPECAN_EXPECTATION5_LEN (25 + 5 * 16 + 1)
/* vanilla batches (VanillaHmm, impl/continuousHmm.c:373-466): skip bins 0..29 beta (match -> gapX), 30..59
 * alpha (gapX -> gapX) (cell_signal_updateBetaAndAlphaProb :478-498) + likelihood, per modelsv_create id */
#define CPECAN_EXPECTATIONV_LEN (60 + 1)
/* HDP batches (HdpHmm, impl/continuousHmm.c:631-697): 9 transitions [from*3+to] + likelihood per
 * cpecan_hip_modelsh_create id; params->threshold is the HdpHmm's assignment threshold, and the event-to-k-mer
 * assignments of an item (cell_signal_updateTransAndKmerSkipExpectations2 impl/pairwiseAligner.c:445-476) come
 * back through cpecan_hip_batch_fetch_pairs as (from state, x, y) triples in the reference's order, logp = the
 * transition's log posterior */
#define CPECAN_EXPECTATIONH_LEN (9 + 1)
int cpecan_hip_batch_expectations_device_ptr(cpecan_batch *batch, void **dev_ptr, int64_t *n_doubles);
int cpecan_hip_batch_fetch_expectations(cpecan_batch *batch, int32_t model_id, double *out);
/* Debug: forward cells and backward cells (as they stand when posteriors are taken) of an item,
 * [cell][state] with cells ordered by diagonal then x-y; needs CPECAN_FLAG_DEBUG_DUMP. */
int cpecan_hip_batch_debug_cells(cpecan_batch *batch, int64_t item, double *forward,
                                 double *backward, int64_t n_cells);
int cpecan_hip_batch_destroy(cpecan_batch *batch);

/* Self-test of the systolic kernel's division: evaluates (x - mu) / sigma for n pseudo-random operand
 * triples (event-like x, model-like mu and sigma, derived from seed) both as an IEEE division and as
 * the Markstein-corrected multiply by RN(1/sigma) the kernel uses; *mismatches receives the number
 * of operands where the two doubles differ (expected: 0). */
int cpecan_hip_selftest_division(cpecan_ctx *ctx, int64_t n, uint64_t seed, int64_t *mismatches);

/* Device and pinned host memory released by batches and model tables is kept by the library for the next request
 * (hipMalloc / hipFree wait for the device; up to half of the card's memory by default, CPECAN_ALLOC_CACHE_GB /
 * CPECAN_PINNED_CACHE_GB).  This gives all of it back to the runtime: call it between phases when something else in the
 * process, or on the card, needs the memory. */
int cpecan_hip_trim_cache(void);

/* Stream of the context as an opaque pointer (a hipStream_t) for callers that need to order
 * their own work (e.g. an RCCL all-reduce of the expectations) after the batch kernels. */
int cpecan_hip_ctx_stream(cpecan_ctx *ctx, void **stream);

#ifdef __cplusplus
}
#endif
#endif
