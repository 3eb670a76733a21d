/*
 * cpecan_oracle.c -- CPU restatement of cPecan's banded pair-HMM forward/backward/posterior DP.
 *
 * TEST INFRASTRUCTURE ONLY (see cpecan_oracle.h).  Plain C99, one thread, same operation order as
 * the reference so that results are bit-identical to it on the same libm:
 *   - no FMA contraction (build with -ffp-contract=off), float-suffixed literals kept as floats,
 *   - the backward pass is the reference's *scatter* into the two earlier diagonals,
 *   - totalProbability is refreshed on the reference's schedule (every 10th posterior diagonal of
 *     a traceback window) with its sequential logAdd fold.
 * Data structures are the build's own (flat arrays, no sonLib); citations name the reference lines
 * each routine follows.
 */
#include "cpecan_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LOG_ZERO (-INFINITY)

/* ------------------------------------------------------------------------------------------ */
/* logAdd: impl/pairwiseAligner.c:235-255                                                     */
/* ------------------------------------------------------------------------------------------ */
static inline double orc_lookup(double x) {
    /* four cubic pieces; coefficients are float literals promoted to double (:242-248) */
    if (x <= 1.00f)
        return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x
               + 0.693203116424741f;
    if (x <= 2.50f)
        return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x
               + 0.692140569840976f;
    if (x <= 4.50f)
        return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x
               + 0.514272634594009f;
    return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x
           + 0.168037164329057f;
}

double orc_logAdd(double x, double y) {
    if (x < y) return (x == LOG_ZERO || y - x >= 7.5) ? y : orc_lookup(y - x) + x;
    return (y == LOG_ZERO || x - y >= 7.5) ? x : orc_lookup(x - y) + y;
}

/* ------------------------------------------------------------------------------------------ */
/* Emissions                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* emissions_discrete_getBaseIndex impl/stateMachine.c:104-118 */
static inline int64_t base_index(char b) {
    switch (b) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return ORC_NUM_KMERS + 1;
    }
}

/* emissions_discrete_getKmerIndex (:120-139) applied to the 6-char copy the emission functions make
 * (:177-181,:602-606): most-significant-first base-4 positional value; any non-ACGT character
 * contributes 4097*weight so the index exceeds NUM_OF_KMERS. A NUL inside the 6 characters
 * shortens strlen() in the reference; the index is then > 4096 too (first char 'n' or the NUL
 * itself is non-ACGT), which is all callers test for. */
int64_t orc_kmer_index(const char *kmer) {
    int64_t x = 0, l = ORC_NUM_KMERS / 4;
    for (int i = 0; i < ORC_KMER_LEN - 1; i++) {
        x += l * base_index(kmer[i]);
        l /= 4;
    }
    x += base_index(kmer[ORC_KMER_LEN - 1]);
    return x;
}

/* emissions_signal_logGaussPdf :333-343 */
double orc_logGaussPdf(double x, double mu, double sigma) {
    if (sigma == 0.0) return LOG_ZERO;
    double log_inv_sqrt_2pi = -0.91893853320467267;
    double l_sigma = log(sigma);
    double a = (x - mu) / sigma;
    return log_inv_sqrt_2pi - l_sigma + (-0.5 * a * a);
}

/* model accessors :221-240: index > NUM_OF_KMERS reads as 0.0 */
static inline double model_get(const double *model, int64_t k, int j) {
    return k > ORC_NUM_KMERS ? 0.0 : model[1 + (k * ORC_MODEL_PARAMS + j)];
}

/* emissions_signal_strawManGetKmerEventMatchProb :595-629 */
double orc_strawman_match(const double *model, int64_t k, const double *event) {
    double eventMean = event[0];
    double eventNoise = event[1];
    double levelMean = model_get(model, k, 0);
    double levelStdDev = model_get(model, k, 1);
    double noiseMean = model_get(model, k, 2);
    double noiseStdDev = model_get(model, k, 3);
    double l_probEventMean = orc_logGaussPdf(eventMean, levelMean, levelStdDev);
    double l_probEventNoise = orc_logGaussPdf(eventNoise, noiseMean, noiseStdDev);
    return l_probEventMean + l_probEventNoise;
}

/* emissions_signal_logInvGaussPdf :322-331 */
static inline double log_inv_gauss(double eventNoise, double modelNoiseMean, double modelNoiseLambda) {
    double l_twoPi = 1.8378770664093453;
    double l_eventNoise = log(eventNoise);
    double a = (eventNoise - modelNoiseMean) / modelNoiseMean;
    double l_modelNoseLambda = log(modelNoiseLambda);
    return (l_modelNoseLambda - l_twoPi - 3 * l_eventNoise - modelNoiseLambda * a * a / eventNoise) / 2;
}

/* emissions_signal_getEventMatchProbWithTwoDists :499-528 for k-mer index k */
double orc_vanilla_match(const double *model, int64_t k, const double *event) {
    double levelProb = orc_logGaussPdf(event[0], model_get(model, k, 0), model_get(model, k, 1));
    double noiseProb = log_inv_gauss(event[1], model_get(model, k, 2), model_get(model, k, 4));
    return levelProb + noiseProb;
}

/* emissions_signal_getKmerSkipBin :388-419 */
static inline int64_t skip_bin(const double *matchModel, int64_t k_im1, int64_t k_i) {
    double d = fabs(model_get(matchModel, k_i, 0) - model_get(matchModel, k_im1, 0));
    int64_t bin = (int64_t) (d / 0.5);
    return bin >= 30 ? 29 : bin;
}

/* grid_spline_interp impl/hdp_math_utils.c:471-495 (evenly spaced grid) */
double orc_grid_spline_interp(double query_x, const double *x, const double *y, const double *slope,
                              int64_t length) {
    if (query_x <= x[0]) return y[0] - slope[0] * (x[0] - query_x);
    if (query_x >= x[length - 1]) {
        int64_t n = length - 1;
        return y[n] + slope[n] * (query_x - x[n]);
    }
    double dx = x[1] - x[0];
    int64_t idx_left = (int64_t) ((query_x - x[0]) / dx);
    int64_t idx_right = idx_left + 1;
    double dy = y[idx_right] - y[idx_left];
    double a = slope[idx_left] * dx - dy;
    double b = dy - slope[idx_right] * dx;
    double t_left = (query_x - x[idx_left]) / dx;
    double t_right = 1.0 - t_left;
    return t_right * y[idx_left] + t_left * y[idx_right] + t_left * t_right * (a * t_right + b * t_left);
}

/* kmer_id impl/nanopore_hdp.c:348-380: digits over the (sorted) alphabet, most significant first */
int64_t orc_hdp_kmer_id(const orc_model *m, const char *kmer) {
    int64_t id = 0;
    for (int i = 0; i < ORC_KMER_LEN; i++) {
        int j = 0;
        while (j < m->alphabetSize && kmer[i] != m->alphabet[j]) j++;
        if (j == m->alphabetSize) return -1; /* the reference exits here (:364-367) */
        id = id * m->alphabetSize + j;
    }
    return id;
}

/* get_nanopore_kmer_density :390 -> dir_proc_density impl/hdp.c:2577-2601 (the walk to the nearest
 * observed ancestor is folded into hdpRow) */
double orc_hdp_density(const orc_model *m, const char *kmer, double x) {
    int64_t id = orc_hdp_kmer_id(m, kmer);
    if (id < 0) return NAN;
    int64_t row = m->hdpRow[id];
    double interp = orc_grid_spline_interp(x, m->hdpGrid, m->hdpY + row * m->gridLength,
                                           m->hdpSlope + row * m->gridLength, m->gridLength);
    return interp > 0.0 ? interp : 0.0;
}

/* emissions_kmer_getGapProb :175-187 */
double orc_kmer_gap(const double *gapX, int64_t k) {
    return k > ORC_NUM_KMERS ? LOG_ZERO : gapX[k];
}

/* emissions_signal_scaleModel :631-651 (match table only) */
void orc_scale_model(double *m, double scale, double shift, double var, double scale_sd,
                     double var_sd) {
    for (int64_t i = 1; i < (ORC_NUM_KMERS * ORC_MODEL_PARAMS) + 1; i += ORC_MODEL_PARAMS) {
        m[i] = m[i] * scale + shift;
        m[i + 1] = m[i + 1] * var;
        m[i + 2] = m[i + 2] * scale_sd;
        m[i + 4] = m[i + 4] * var_sd;
        m[i + 3] = sqrt(pow(m[i + 2], 3.0) / m[i + 4]);
    }
}

void orc_defaults_sm3_nanopore(orc_model *m) {
    /* stateMachine3_setTransitionsToNanoporeDefaults :1278-1289 */
    m->kind = ORC_SM3_STRAWMAN;
    m->stateNumber = 3;
    m->t[ORC_T3_MATCH_CONTINUE] = -0.23552123624314988;
    m->t[ORC_T3_MATCH_FROM_GAP_X] = -0.21880828092192281;
    m->t[ORC_T3_MATCH_FROM_GAP_Y] = -0.013406326748077823;
    m->t[ORC_T3_GAP_OPEN_X] = -1.6269694202638481;
    m->t[ORC_T3_GAP_OPEN_Y] = -4.3187242127300092;
    m->t[ORC_T3_GAP_EXTEND_X] = -1.6269694202638481;
    m->t[ORC_T3_GAP_EXTEND_Y] = -4.3187242127239411;
    m->t[ORC_T3_GAP_SWITCH_TO_X] = LOG_ZERO;
    m->t[ORC_T3_GAP_SWITCH_TO_Y] = LOG_ZERO;
}

/* sm5 transition slots follow struct _StateMachine5 (inc/stateMachine.h:108-124) */
enum { /* _StateMachine4's members in order (inc/stateMachine.h:134-152) */
    T4_MATCH_CONTINUE = 0, T4_MATCH_FROM_SHORT_GAP_X, T4_MATCH_FROM_LONG_GAP_X, T4_MATCH_FROM_SHORT_GAP_Y,
    T4_GAP_SHORT_OPEN_X, T4_GAP_SHORT_EXTEND_X, T4_GAP_SHORT_OPEN_Y, T4_GAP_SHORT_EXTEND_Y,
    T4_GAP_LONG_OPEN_X, T4_GAP_LONG_EXTEND_X, T4_GAP_LONG_SWITCH_TO_X
};
enum {
    T5_MATCH_CONTINUE = 0, T5_MATCH_FROM_SHORT_GAP_X, T5_MATCH_FROM_LONG_GAP_X, T5_GAP_SHORT_OPEN_X,
    T5_GAP_SHORT_EXTEND_X, T5_GAP_SHORT_SWITCH_TO_X, T5_GAP_LONG_OPEN_X, T5_GAP_LONG_EXTEND_X,
    T5_GAP_LONG_SWITCH_TO_X, T5_MATCH_FROM_SHORT_GAP_Y, T5_MATCH_FROM_LONG_GAP_Y,
    T5_GAP_SHORT_OPEN_Y, T5_GAP_SHORT_EXTEND_Y, T5_GAP_SHORT_SWITCH_TO_Y, T5_GAP_LONG_OPEN_Y,
    T5_GAP_LONG_EXTEND_Y, T5_GAP_LONG_SWITCH_TO_Y
};

void orc_defaults_hdp(orc_model *m) {
    orc_defaults_sm3_nanopore(m); /* getHdpStateMachine3 :1738: stateMachine3_setTransitionsToNanoporeDefaults */
    m->kind = ORC_SM3_HDP;
    m->stateNumber = 3;
}

void orc_defaults_vanilla(orc_model *m) {
    m->kind = ORC_SM3_VANILLA;
    m->stateNumber = 3;
    m->t[0] = 0.17;                 /* TRANSITION_M_TO_Y_NOT_X (:1575) */
    m->t[1] = 0.55f;                /* TRANSITION_E_TO_E, a float literal (:1576) */
    m->t[2] = -0.23552123624314988; /* DEFAULT_END_MATCH_PROB  */
    m->t[3] = -1.6269694202638481;  /* DEFAULT_END_FROM_X_PROB */
    m->t[4] = -4.3187242127300092;  /* DEFAULT_END_FROM_Y_PROB */
}

void orc_defaults_sm4(orc_model *m) {
    /* impl/stateMachine.c:992-1011: "set transitions to defaults (these are from a template read)" */
    m->kind = ORC_SM4_SIGNAL;
    m->stateNumber = 4;
    m->t[T4_MATCH_CONTINUE] = -0.23552123624314988;
    m->t[T4_GAP_SHORT_OPEN_X] = -1.6269694202638481;
    m->t[T4_GAP_SHORT_OPEN_Y] = -4.7241893208381773;
    m->t[T4_GAP_LONG_OPEN_X] = -5.4173365013981227;
    m->t[T4_GAP_SHORT_EXTEND_X] = -1.6269694202638481;
    m->t[T4_MATCH_FROM_SHORT_GAP_X] = -0.21880828092192281;
    m->t[T4_GAP_LONG_EXTEND_X] = -0.003442492794189331;
    m->t[T4_MATCH_FROM_LONG_GAP_X] = -5.6732801731704612;
    m->t[T4_MATCH_FROM_SHORT_GAP_Y] = -0.013406326748077823;
    m->t[T4_GAP_SHORT_EXTEND_Y] = -4.724189320832104;
    m->t[T4_GAP_LONG_SWITCH_TO_X] = -5.4173365013920494;
}

void orc_defaults_sm5(orc_model *m, double *match16, double *gap4x, double *gap4y) {
    /* stateMachine5_construct :920-937 and emissions_symbol_setEmissionsToDefaults :60-82 */
    m->kind = ORC_SM5_SYMBOL;
    m->stateNumber = 5;
    m->t[T5_MATCH_CONTINUE] = -0.030064059121770816;
    m->t[T5_MATCH_FROM_SHORT_GAP_X] = -1.272871422049609;
    m->t[T5_MATCH_FROM_LONG_GAP_X] = -5.673280173170473;
    m->t[T5_GAP_SHORT_OPEN_X] = -4.34381910900448;
    m->t[T5_GAP_SHORT_EXTEND_X] = -0.3388262689231553;
    m->t[T5_GAP_SHORT_SWITCH_TO_X] = -4.910694825551255;
    m->t[T5_GAP_LONG_OPEN_X] = -6.30810595366929;
    m->t[T5_GAP_LONG_EXTEND_X] = -0.003442492794189331;
    m->t[T5_GAP_LONG_SWITCH_TO_X] = -6.30810595366929;
    m->t[T5_MATCH_FROM_SHORT_GAP_Y] = m->t[T5_MATCH_FROM_SHORT_GAP_X];
    m->t[T5_MATCH_FROM_LONG_GAP_Y] = m->t[T5_MATCH_FROM_LONG_GAP_X];
    m->t[T5_GAP_SHORT_OPEN_Y] = m->t[T5_GAP_SHORT_OPEN_X];
    m->t[T5_GAP_SHORT_EXTEND_Y] = m->t[T5_GAP_SHORT_EXTEND_X];
    m->t[T5_GAP_SHORT_SWITCH_TO_Y] = m->t[T5_GAP_SHORT_SWITCH_TO_X];
    m->t[T5_GAP_LONG_OPEN_Y] = m->t[T5_GAP_LONG_OPEN_X];
    m->t[T5_GAP_LONG_EXTEND_Y] = m->t[T5_GAP_LONG_EXTEND_X];
    m->t[T5_GAP_LONG_SWITCH_TO_Y] = m->t[T5_GAP_LONG_SWITCH_TO_X];
    const double EM = -2.1149196655034745, ETV = -4.5691014376830479, ETS = -3.9833860032220842;
    const double M[16] = { EM, ETV, ETS, ETV, ETV, EM, ETV, ETS, ETS, ETV, EM, ETV, ETV, ETS, ETV, EM };
    memcpy(match16, M, sizeof(M));
    for (int i = 0; i < 4; i++) gap4x[i] = gap4y[i] = -1.6094379124341003;
    m->match = match16;
    m->gapX = gap4x;
    m->gapY = gap4y;
}

void orc_params_default(orc_params *p) {
    p->threshold = 0.01;
    p->minDiagsBetweenTraceBack = 1000;
    p->traceBackDiagonals = 40;
    p->diagonalExpansion = 20;
    p->splitMatrixBiggerThanThis = (int64_t) 3000 * 3000;
}

/* ------------------------------------------------------------------------------------------ */
/* Sequences: sequence_getKmer / sequence_getBase / sequence_getEvent  :308-337                */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const orc_model *m;
    const char *x;      /* nucleotides */
    int64_t lX;
    const void *y;      /* events (sm3) or nucleotides (sm5) */
    int64_t lY;
} seqs_t;

static const double NULLEVENT[2] = { LOG_ZERO, 0 }; /* :261 */

/* per-cell symbols handed to the cell function */
typedef struct {
    int64_t kx;        /* sm3: k-mer index of X element (>4096 invalid); sm5: base index */
    int64_t ky;        /* sm5: base index of Y */
    int64_t kp;        /* vanilla: index of the k-mer before kx (sequence_getKmer2) */
    const char *kmer;  /* HDP: the k-mer's characters (sequence_getKmer3) */
    const double *ev;  /* sm3: event */
} symbols_t;

static inline void get_symbols(const seqs_t *s, int64_t ix, int64_t iy, symbols_t *o) {
    o->kp = 0;
    o->kmer = NULL;
    if (s->m->kind == ORC_SM3_HDP) {
        /* sequence_getKmer3 (:327-331): index < 0 reads the first k-mer */
        o->kmer = s->x + (ix >= 0 ? ix : 0);
        o->kx = 0;
        o->ev = iy >= 0 ? ((const double *) s->y) + 3 * iy : NULLEVENT;
        o->ky = 0;
    } else if (s->m->kind == ORC_SM3_VANILLA) {
        /* sequence_getKmer2 (:320-325): a pointer to char max(ix-1, 0); the emission code reads the
         * k-mer at +1 and the skip-bin code the k-mers at +0 and +1 (so element 0 is scored as k-mer 1) */
        const char *p = s->x + (ix > 0 ? ix - 1 : 0);
        o->kp = orc_kmer_index(p);
        o->kx = orc_kmer_index(p + 1);
        o->ev = iy >= 0 ? ((const double *) s->y) + 3 * iy : NULLEVENT;
        o->ky = 0;
    } else if (s->m->kind == ORC_SM3_STRAWMAN || s->m->kind == ORC_SM4_SIGNAL) {
        /* index < 0 yields the literal "n" (:315-317): first char non-ACGT => index > 4096 */
        o->kx = ix >= 0 ? orc_kmer_index(s->x + ix) : (int64_t) ORC_NUM_KMERS * 4097;
        o->ev = iy >= 0 ? ((const double *) s->y) + 3 * iy : NULLEVENT;
        o->ky = 0;
    } else {
        o->kx = ix >= 0 ? base_index(s->x[ix]) : ORC_NUM_KMERS + 1;
        o->ky = iy >= 0 ? base_index(((const char *) s->y)[iy]) : ORC_NUM_KMERS + 1;
        o->ev = NULL;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Cell recurrences                                                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    double total;           /* totalProbability of the enclosing diagonal */
    orc_expectations *hmm;  /* an orc_expectations5 for the 5-state symbol machine */
    int64_t kx, ky, kp;
    int64_t ix, iy;         /* sequence indices of the cell's elements */
} exp_args_t;

typedef void (*trans_fn)(double *from, double *to, int f, int t, double eP, double tP, void *extra);

/* doTransitionForward :365-370 */
static void trans_forward(double *from, double *to, int f, int t, double eP, double tP, void *e) {
    (void) e;
    to[t] = orc_logAdd(to[t], from[f] + (eP + tP));
}

/* doTransitionBackward :378-383 */
static void trans_backward(double *from, double *to, int f, int t, double eP, double tP, void *e) {
    (void) e;
    from[f] = orc_logAdd(from[f], to[t] + (eP + tP));
}

/* cell_signal_updateTransAndKmerSkipExpectations :426-443 */
static void trans_expect_sm3(double *from, double *to, int f, int t, double eP, double tP, void *e) {
    exp_args_t *a = (exp_args_t *) e;
    double p = exp(from[f] + to[t] + (eP + tP) - a->total);
    a->hmm->transitions[f * 3 + t] += p;
    if (t == 1) { /* shortGapX */
        if (a->kx >= 0 && a->kx < ORC_NUM_KMERS) /* reference writes out of bounds otherwise */
            a->hmm->kmerGap[a->kx] += p;
    }
}

/* cell_signal_updateBetaAndAlphaProb :478-498: the skip bin of the cell's k-mer pair (taken on the hmm's
 * copy of the match model, vanillaHmm_implantMatchModelsintoHmm) collects match->gapX in bin and
 * gapX->gapX in bin + 30 */
static const double *g_vanilla_match; /* the running model's match table (test infrastructure: one thread) */
static void trans_expect_vanilla(double *from, double *to, int f, int t, double eP, double tP, void *e) {
    exp_args_t *a = (exp_args_t *) e;
    orc_expectations_v *h = (orc_expectations_v *) a->hmm;
    const int64_t bin = skip_bin(g_vanilla_match, a->kp, a->kx);
    double p = exp(from[f] + to[t] + (eP + tP) - a->total);
    if (f == 0 && t == 1) h->kmerSkipBins[bin] += p;
    if (f == 1 && t == 1) h->kmerSkipBins[bin + 30] += p;
}

/* cell_signal_updateTransAndKmerSkipExpectations2 :445-476 */
static void trans_expect_hdp(double *from, double *to, int f, int t, double eP, double tP, void *e) {
    exp_args_t *a = (exp_args_t *) e;
    orc_expectations_h *h = (orc_expectations_h *) a->hmm;
    const double lp = from[f] + to[t] + (eP + tP) - a->total;
    const double p = exp(lp);
    h->transitions[f * 3 + t] += p;
    if (t == 0 && p >= h->threshold) {
        if (h->n == h->cap) {
            h->cap = h->cap ? 2 * h->cap : 256;
            h->assign = realloc(h->assign, sizeof(int64_t) * 3 * (size_t) h->cap);
            h->logp = realloc(h->logp, sizeof(double) * (size_t) h->cap);
        }
        h->assign[3 * h->n] = f;
        h->assign[3 * h->n + 1] = a->ix;
        h->assign[3 * h->n + 2] = a->iy;
        h->logp[h->n++] = lp;
    }
}

/* cell_updateExpectations :407-424 (HmmDiscrete: transitions [from*5+to], emissions [to][x][y]) */
static void trans_expect_sm5(double *from, double *to, int f, int t, double eP, double tP, void *e) {
    exp_args_t *a = (exp_args_t *) e;
    orc_expectations5 *h = (orc_expectations5 *) a->hmm;
    double p = exp(from[f] + to[t] + (eP + tP) - a->total);
    h->transitions[f * 5 + t] += p;
    if (a->kx < 4 && a->ky < 4) /* ignore gaps involving Ns */
        h->emissions[t * 16 + a->kx * 4 + a->ky] += p;
}

enum { ST_MATCH = 0, ST_SHORT_GAP_X = 1, ST_SHORT_GAP_Y = 2, ST_LONG_GAP_X = 3, ST_LONG_GAP_Y = 4 };

/* stateMachine3_cellCalculate impl/stateMachine.c:1305-1334 */
static void cell_sm3(const orc_model *m, double *cur, double *lower, double *middle, double *upper,
                     const symbols_t *s, trans_fn fn, void *extra) {
    const double *t = m->t;
    if (lower != NULL) {
        double eP = orc_kmer_gap(m->gapX, s->kx);
        fn(lower, cur, ST_MATCH, ST_SHORT_GAP_X, eP, t[ORC_T3_GAP_OPEN_X], extra);
        fn(lower, cur, ST_SHORT_GAP_X, ST_SHORT_GAP_X, eP, t[ORC_T3_GAP_EXTEND_X], extra);
        fn(lower, cur, ST_SHORT_GAP_Y, ST_SHORT_GAP_X, eP, t[ORC_T3_GAP_SWITCH_TO_X], extra);
    }
    if (middle != NULL) {
        double eP = orc_strawman_match(m->match, s->kx, s->ev);
        fn(middle, cur, ST_MATCH, ST_MATCH, eP, t[ORC_T3_MATCH_CONTINUE], extra);
        fn(middle, cur, ST_SHORT_GAP_X, ST_MATCH, eP, t[ORC_T3_MATCH_FROM_GAP_X], extra);
        fn(middle, cur, ST_SHORT_GAP_Y, ST_MATCH, eP, t[ORC_T3_MATCH_FROM_GAP_Y], extra);
    }
    if (upper != NULL) {
        double eP = orc_strawman_match(m->gapY, s->kx, s->ev);
        fn(upper, cur, ST_MATCH, ST_SHORT_GAP_Y, eP, t[ORC_T3_GAP_OPEN_Y], extra);
        fn(upper, cur, ST_SHORT_GAP_Y, ST_SHORT_GAP_Y, eP, t[ORC_T3_GAP_EXTEND_Y], extra);
    }
}

/* stateMachine4_cellCalculate impl/stateMachine.c:867-897 */
static void cell_sm4(const orc_model *m, double *cur, double *lower, double *middle, double *upper,
                     const symbols_t *s, trans_fn fn, void *extra) {
    const double *t = m->t;
    if (lower != NULL) {
        double eP = orc_kmer_gap(m->gapX, s->kx);
        fn(lower, cur, ST_MATCH, ST_SHORT_GAP_X, eP, t[T4_GAP_SHORT_OPEN_X], extra);
        fn(lower, cur, ST_SHORT_GAP_X, ST_SHORT_GAP_X, eP, t[T4_GAP_SHORT_EXTEND_X], extra);
        fn(lower, cur, ST_MATCH, ST_LONG_GAP_X, eP, t[T4_GAP_LONG_OPEN_X], extra);
        fn(lower, cur, ST_LONG_GAP_X, ST_LONG_GAP_X, eP, t[T4_GAP_LONG_EXTEND_X], extra);
        fn(lower, cur, ST_SHORT_GAP_Y, ST_LONG_GAP_X, eP, t[T4_GAP_LONG_SWITCH_TO_X], extra);
    }
    if (middle != NULL) {
        double eP = orc_strawman_match(m->match, s->kx, s->ev);
        fn(middle, cur, ST_MATCH, ST_MATCH, eP, t[T4_MATCH_CONTINUE], extra);
        fn(middle, cur, ST_SHORT_GAP_X, ST_MATCH, eP, t[T4_MATCH_FROM_SHORT_GAP_X], extra);
        fn(middle, cur, ST_SHORT_GAP_Y, ST_MATCH, eP, t[T4_MATCH_FROM_SHORT_GAP_Y], extra);
        fn(middle, cur, ST_LONG_GAP_X, ST_MATCH, eP, t[T4_MATCH_FROM_LONG_GAP_X], extra);
    }
    if (upper != NULL) {
        double eP = orc_strawman_match(m->gapY, s->kx, s->ev);
        fn(upper, cur, ST_MATCH, ST_SHORT_GAP_Y, eP, t[T4_GAP_SHORT_OPEN_Y], extra);
        fn(upper, cur, ST_SHORT_GAP_Y, ST_SHORT_GAP_Y, eP, t[T4_GAP_SHORT_EXTEND_Y], extra);
    }
}

/* stateMachine3HDP_cellCalculate impl/stateMachine.c:1336-1366 */
static void cell_hdp(const orc_model *m, double *cur, double *lower, double *middle, double *upper,
                     const symbols_t *s, trans_fn fn, void *extra) {
    const double *t = m->t;
    if (lower != NULL) {
        double eP = -2.3025850929940455; /* log(0.1) */
        fn(lower, cur, ST_MATCH, ST_SHORT_GAP_X, eP, t[ORC_T3_GAP_OPEN_X], extra);
        fn(lower, cur, ST_SHORT_GAP_X, ST_SHORT_GAP_X, eP, t[ORC_T3_GAP_EXTEND_X], extra);
        fn(lower, cur, ST_SHORT_GAP_Y, ST_SHORT_GAP_X, eP, t[ORC_T3_GAP_SWITCH_TO_X], extra);
    }
    if (middle != NULL) {
        double eP = orc_hdp_density(m, s->kmer, s->ev[0]);
        fn(middle, cur, ST_MATCH, ST_MATCH, eP, t[ORC_T3_MATCH_CONTINUE], extra);
        fn(middle, cur, ST_SHORT_GAP_X, ST_MATCH, eP, t[ORC_T3_MATCH_FROM_GAP_X], extra);
        fn(middle, cur, ST_SHORT_GAP_Y, ST_MATCH, eP, t[ORC_T3_MATCH_FROM_GAP_Y], extra);
    }
    if (upper != NULL) {
        double eP = orc_hdp_density(m, s->kmer, s->ev[0]);
        fn(upper, cur, ST_MATCH, ST_SHORT_GAP_Y, eP, t[ORC_T3_GAP_OPEN_Y], extra);
        fn(upper, cur, ST_SHORT_GAP_Y, ST_SHORT_GAP_Y, eP, t[ORC_T3_GAP_EXTEND_Y], extra);
    }
}

/* stateMachine3Vanilla_cellCalculate impl/stateMachine.c:1368-1409 */
static void cell_vanilla(const orc_model *m, double *cur, double *lower, double *middle, double *upper,
                         const symbols_t *s, trans_fn fn, void *extra) {
    const int64_t bin = skip_bin(m->match, s->kp, s->kx);
    double a_mx = m->gapX[bin];      /* beta  */
    double a_my = (1 - a_mx) * m->t[0];
    double a_mm = 1.0f - a_my - a_mx;
    double a_yy = m->t[1];
    double a_ym = 1.0f - a_yy;
    double a_xx = m->gapX[bin + 30]; /* alpha */
    double a_xm = 1.0f - a_xx;
    if (lower != NULL) {
        fn(lower, cur, ST_MATCH, ST_SHORT_GAP_X, 0, log(a_mx), extra);
        fn(lower, cur, ST_SHORT_GAP_X, ST_SHORT_GAP_X, 0, log(a_xx), extra);
    }
    if (middle != NULL) {
        double eP = orc_vanilla_match(m->match, s->kx, s->ev);
        fn(middle, cur, ST_MATCH, ST_MATCH, eP, log(a_mm), extra);
        fn(middle, cur, ST_SHORT_GAP_X, ST_MATCH, eP, log(a_xm), extra);
        fn(middle, cur, ST_SHORT_GAP_Y, ST_MATCH, eP, log(a_ym), extra);
    }
    if (upper != NULL) {
        double eP = orc_vanilla_match(m->gapY, s->kx, s->ev);
        fn(upper, cur, ST_MATCH, ST_SHORT_GAP_Y, eP, log(a_my), extra);
        fn(upper, cur, ST_SHORT_GAP_Y, ST_SHORT_GAP_Y, eP, log(a_yy), extra);
    }
}

/* emissions_symbol_getGapProb / getMatchProb :155-173; the i==4 branches never fire because the
 * base index of N is 4097 (quirk Q3): N-free input is a precondition here. */
static inline double sym_gap(const double *g, int64_t i) { return i < 4 ? g[i] : LOG_ZERO; }
static inline double sym_match(const double *mm, int64_t ix, int64_t iy) {
    return (ix < 4 && iy < 4) ? mm[ix * 4 + iy] : LOG_ZERO;
}

/* stateMachine5_cellCalculate impl/stateMachine.c:829-865 */
static void cell_sm5(const orc_model *m, double *cur, double *lower, double *middle, double *upper,
                     const symbols_t *s, trans_fn fn, void *extra) {
    const double *t = m->t;
    if (lower != NULL) {
        double eP = sym_gap(m->gapX, s->kx);
        fn(lower, cur, ST_MATCH, ST_SHORT_GAP_X, eP, t[T5_GAP_SHORT_OPEN_X], extra);
        fn(lower, cur, ST_SHORT_GAP_X, ST_SHORT_GAP_X, eP, t[T5_GAP_SHORT_EXTEND_X], extra);
        fn(lower, cur, ST_MATCH, ST_LONG_GAP_X, eP, t[T5_GAP_LONG_OPEN_X], extra);
        fn(lower, cur, ST_LONG_GAP_X, ST_LONG_GAP_X, eP, t[T5_GAP_LONG_EXTEND_X], extra);
    }
    if (middle != NULL) {
        double eP = sym_match(m->match, s->kx, s->ky);
        fn(middle, cur, ST_MATCH, ST_MATCH, eP, t[T5_MATCH_CONTINUE], extra);
        fn(middle, cur, ST_SHORT_GAP_X, ST_MATCH, eP, t[T5_MATCH_FROM_SHORT_GAP_X], extra);
        fn(middle, cur, ST_SHORT_GAP_Y, ST_MATCH, eP, t[T5_MATCH_FROM_SHORT_GAP_Y], extra);
        fn(middle, cur, ST_LONG_GAP_X, ST_MATCH, eP, t[T5_MATCH_FROM_LONG_GAP_X], extra);
        fn(middle, cur, ST_LONG_GAP_Y, ST_MATCH, eP, t[T5_MATCH_FROM_LONG_GAP_Y], extra);
    }
    if (upper != NULL) {
        double eP = sym_gap(m->gapY, s->ky);
        fn(upper, cur, ST_MATCH, ST_SHORT_GAP_Y, eP, t[T5_GAP_SHORT_OPEN_Y], extra);
        fn(upper, cur, ST_SHORT_GAP_Y, ST_SHORT_GAP_Y, eP, t[T5_GAP_SHORT_EXTEND_Y], extra);
        fn(upper, cur, ST_MATCH, ST_LONG_GAP_Y, eP, t[T5_GAP_LONG_OPEN_Y], extra);
        fn(upper, cur, ST_LONG_GAP_Y, ST_LONG_GAP_Y, eP, t[T5_GAP_LONG_EXTEND_Y], extra);
    }
}

static inline void cell_calc(const orc_model *m, double *cur, double *lower, double *middle,
                             double *upper, const symbols_t *s, trans_fn fn, void *extra) {
    if (m->kind == ORC_SM3_HDP) cell_hdp(m, cur, lower, middle, upper, s, fn, extra);
    else if (m->kind == ORC_SM3_VANILLA) cell_vanilla(m, cur, lower, middle, upper, s, fn, extra);
    else if (m->kind == ORC_SM3_STRAWMAN) cell_sm3(m, cur, lower, middle, upper, s, fn, extra);
    else if (m->kind == ORC_SM4_SIGNAL) cell_sm4(m, cur, lower, middle, upper, s, fn, extra);
    else cell_sm5(m, cur, lower, middle, upper, s, fn, extra);
}

/* start / end state vectors: impl/stateMachine.c:1168-1207 (sm3), :743-789 (sm5) */
static double state_value(const orc_model *m, int which, int s) {
    /* which: 0 start, 1 raggedStart, 2 end, 3 raggedEnd */
    const double *t = m->t;
    if (m->kind == ORC_SM3_VANILLA) { /* start as sm3; end :1209-1235 */
        switch (which) {
        case 0: return s == ST_MATCH ? 0 : LOG_ZERO;
        case 1: return (s == ST_SHORT_GAP_X || s == ST_SHORT_GAP_Y) ? 0 : LOG_ZERO;
        case 2: return s == ST_MATCH ? t[2] : s == ST_SHORT_GAP_X ? t[3] : t[4];
        default: return s == ST_MATCH ? (t[3] + t[4]) / 2.0 : s == ST_SHORT_GAP_X ? t[3] : t[4];
        }
    }
    if (m->kind == ORC_SM4_SIGNAL) { /* start: stateMachine5's (:743); the rest impl/stateMachine.c:791-829 */
        switch (which) {
        case 0: return s == ST_MATCH ? 0 : LOG_ZERO;
        case 1: return (s == ST_LONG_GAP_X || s == ST_SHORT_GAP_Y) ? 0 : LOG_ZERO;
        case 2:
            return s == ST_MATCH ? t[T4_MATCH_CONTINUE] : s == ST_SHORT_GAP_X ? t[T4_MATCH_FROM_SHORT_GAP_X]
                   : s == ST_SHORT_GAP_Y ? t[T4_MATCH_FROM_SHORT_GAP_Y] : t[T4_MATCH_FROM_LONG_GAP_X];
        default: return s == ST_LONG_GAP_X ? t[T4_GAP_LONG_EXTEND_X] : t[T4_GAP_LONG_OPEN_X];
        }
    }
    if (m->kind == ORC_SM3_STRAWMAN || m->kind == ORC_SM3_HDP) {
        switch (which) {
        case 0: return s == ST_MATCH ? 0 : LOG_ZERO;
        case 1: return (s == ST_SHORT_GAP_X || s == ST_SHORT_GAP_Y) ? 0 : LOG_ZERO;
        case 2:
            return s == ST_MATCH ? t[ORC_T3_MATCH_CONTINUE]
                   : s == ST_SHORT_GAP_X ? t[ORC_T3_MATCH_FROM_GAP_X] : t[ORC_T3_MATCH_FROM_GAP_Y];
        default:
            return s == ST_MATCH ? (t[ORC_T3_GAP_OPEN_X] + t[ORC_T3_GAP_OPEN_Y]) / 2.0
                   : s == ST_SHORT_GAP_X ? t[ORC_T3_GAP_EXTEND_X] : t[ORC_T3_GAP_EXTEND_Y];
        }
    }
    switch (which) {
    case 0: return s == ST_MATCH ? 0 : LOG_ZERO;
    case 1: return (s == ST_LONG_GAP_X || s == ST_LONG_GAP_Y) ? 0 : LOG_ZERO;
    case 2:
        switch (s) {
        case ST_MATCH: return t[T5_MATCH_CONTINUE];
        case ST_SHORT_GAP_X: return t[T5_MATCH_FROM_SHORT_GAP_X];
        case ST_SHORT_GAP_Y: return t[T5_MATCH_FROM_SHORT_GAP_Y];
        case ST_LONG_GAP_X: return t[T5_MATCH_FROM_LONG_GAP_X];
        default: return t[T5_MATCH_FROM_LONG_GAP_Y];
        }
    default:
        switch (s) {
        case ST_MATCH: return t[T5_GAP_LONG_OPEN_X];
        case ST_SHORT_GAP_X: return t[T5_GAP_LONG_OPEN_X];
        case ST_SHORT_GAP_Y: return t[T5_GAP_LONG_OPEN_Y];
        case ST_LONG_GAP_X: return t[T5_GAP_LONG_EXTEND_X];
        default: return t[T5_GAP_LONG_EXTEND_Y];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Band: impl/pairwiseAligner.c:93-184                                                        */
/* ------------------------------------------------------------------------------------------ */
static inline int64_t diag_x(int64_t xay, int64_t xmy) { return (xay + xmy) / 2; } /* :67-70 */
static inline int64_t diag_y(int64_t xay, int64_t xmy) { return (xay - xmy) / 2; } /* :76-79 */
static inline int64_t bound_coord(int64_t z, int64_t lZ) { return z < 0 ? 0 : (z > lZ ? lZ : z); }
static inline int64_t avoid_off_by_one(int64_t xay, int64_t xmy) {
    return (xay + xmy) % 2 == 0 ? xmy : xmy + 1;
}
static inline void set_diag_p(int64_t *xmy, int64_t i, int64_t j, int64_t k) {
    if (i < j) *xmy += 2 * (j - i) * k;
}

int orc_band(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY, int64_t expansion,
             int64_t *outL, int64_t *outR) {
    int64_t ai = 0, xay = 0, pxay = 0, pxmy = 0, nxay = 0, nxmy = 0;
    int64_t xL = 0, yL = 0, xU = 0, yU = 0;
    while (xay <= lX + lY) {
        /* band_setCurrentDiagonal :108-126 */
        int64_t xmyL = avoid_off_by_one(xay, xL - yL);
        int64_t xmyR = avoid_off_by_one(xay, xU - yU);
        set_diag_p(&xmyL, diag_x(xay, xmyL), xL, 1);
        set_diag_p(&xmyL, yL, diag_y(xay, xmyL), 1);
        set_diag_p(&xmyR, xU, diag_x(xay, xmyR), -1);
        set_diag_p(&xmyR, diag_y(xay, xmyR), yU, -1);
        /* diagonal_construct validity :37 */
        if ((xay + xmyL) % 2 != 0 || (xay + xmyR) % 2 != 0 || xmyL > xmyR) return -1;
        outL[xay] = xmyL;
        outR[xay] = xmyR;
        if (nxay == xay++) {
            pxay = nxay;
            pxmy = nxmy;
            int64_t x = lX, y = lY;
            if (ai < nAnchors) {
                x = anchors[2 * ai] + 1;
                y = anchors[2 * ai + 1] + 1;
                ai++;
            }
            nxay = x + y;
            nxmy = x - y;
            xL = bound_coord(diag_x(pxay, pxmy - expansion), lX);
            yL = bound_coord(diag_y(nxay, nxmy - expansion), lY);
            xU = bound_coord(diag_x(nxay, nxmy + expansion), lX);
            yU = bound_coord(diag_y(pxay, pxmy + expansion), lY);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Split points: impl/pairwiseAligner.c:1289-1340                                             */
/* ------------------------------------------------------------------------------------------ */
static int split_p(int64_t *x1, int64_t *y1, int64_t x2, int64_t y2, int64_t x3, int64_t y3,
                   int64_t *out, int64_t *n, int64_t cap, int64_t maxSize, int skipBlock) {
    int64_t lX2 = x3 - x2, lY2 = y3 - y2;
    int64_t matrixSize = lX2 * lY2;
    if (matrixSize > maxSize) {
        int64_t maxSequenceLength = sqrt(maxSize);
        int64_t hX = lX2 / 2 > maxSequenceLength ? maxSequenceLength : lX2 / 2;
        int64_t hY = lY2 / 2 > maxSequenceLength ? maxSequenceLength : lY2 / 2;
        if (!skipBlock && *n < cap) {
            out[4 * *n] = *x1; out[4 * *n + 1] = *y1; out[4 * *n + 2] = x2 + hX; out[4 * *n + 3] = y2 + hY;
            (*n)++;
        }
        *x1 = x3 - hX;
        *y1 = y3 - hY;
        return 1;
    }
    return 0;
}

int64_t orc_split_points(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY,
                         int64_t maxSize, int raggedLeft, int raggedRight, int64_t *out,
                         int64_t cap) {
    int64_t x1 = 0, y1 = 0, x2 = 0, y2 = 0, n = 0;
    for (int64_t i = 0; i < nAnchors; i++) {
        int64_t x3 = anchors[2 * i], y3 = anchors[2 * i + 1];
        split_p(&x1, &y1, x2, y2, x3, y3, out, &n, cap, maxSize, raggedLeft && i == 0);
        x2 = x3 + 1;
        y2 = y3 + 1;
    }
    if (!split_p(&x1, &y1, x2, y2, lX, lY, out, &n, cap, maxSize, raggedLeft && nAnchors == 0)
        || !raggedRight) {
        if (n < cap) {
            out[4 * n] = x1; out[4 * n + 1] = y1; out[4 * n + 2] = lX; out[4 * n + 3] = lY;
            n++;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* Results                                                                                    */
/* ------------------------------------------------------------------------------------------ */
orc_result *orc_result_new(void) {
    orc_result *r = calloc(1, sizeof(orc_result));
    return r;
}
void orc_result_free(orc_result *r) {
    if (!r) return;
    free(r->triples); free(r->logp); free(r->totalsXay); free(r->totals); free(r);
}
static void result_push(orc_result *r, int64_t p, int64_t x, int64_t y, double lp) {
    if (r->n == r->cap) {
        r->cap = r->cap ? r->cap * 2 : 1024;
        r->triples = realloc(r->triples, sizeof(int64_t) * 3 * r->cap);
        r->logp = realloc(r->logp, sizeof(double) * r->cap);
    }
    r->triples[3 * r->n] = p; r->triples[3 * r->n + 1] = x; r->triples[3 * r->n + 2] = y;
    r->logp[r->n] = lp;
    r->n++;
}
static void result_push_total(orc_result *r, int64_t xay, double t) {
    if (r->nTotals == r->capTotals) {
        r->capTotals = r->capTotals ? r->capTotals * 2 : 256;
        r->totalsXay = realloc(r->totalsXay, sizeof(int64_t) * r->capTotals);
        r->totals = realloc(r->totals, sizeof(double) * r->capTotals);
    }
    r->totalsXay[r->nTotals] = xay; r->totals[r->nTotals] = t; r->nTotals++;
}

/* ------------------------------------------------------------------------------------------ */
/* DP matrices: one heap block per live diagonal, cell-major/state-minor  :521-659             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t nDiag, S;
    const int64_t *L, *R;  /* band */
    double **d;            /* [nDiag+1] */
} dpm_t;

static void dpm_init(dpm_t *m, int64_t nDiag, int64_t S, const int64_t *L, const int64_t *R) {
    m->nDiag = nDiag; m->S = S; m->L = L; m->R = R;
    m->d = calloc(nDiag + 1, sizeof(double *));
}
static inline int64_t width_of(const int64_t *L, const int64_t *R, int64_t xay) {
    return (R[xay] - L[xay]) / 2 + 1;
}
static double *dpm_create(dpm_t *m, int64_t xay) {
    m->d[xay] = malloc(sizeof(double) * m->S * width_of(m->L, m->R, xay));
    return m->d[xay];
}
static void dpm_delete(dpm_t *m, int64_t xay) {
    if (xay >= 0 && xay <= m->nDiag && m->d[xay]) { free(m->d[xay]); m->d[xay] = NULL; }
}
static inline double *dpm_diag(dpm_t *m, int64_t xay) {
    return (xay < 0 || xay > m->nDiag) ? NULL : m->d[xay];
}
/* dpDiagonal_getCell :562-568 */
static inline double *dpm_cell(dpm_t *m, int64_t xay, int64_t xmy) {
    double *dd = dpm_diag(m, xay);
    if (!dd || xmy < m->L[xay] || xmy > m->R[xay]) return NULL;
    return dd + ((xmy - m->L[xay]) / 2) * m->S;
}
static void diag_fill(dpm_t *m, int64_t xay, const orc_model *mod, int which) {
    double *dd = m->d[xay];
    int64_t w = width_of(m->L, m->R, xay);
    for (int64_t c = 0; c < w; c++)
        for (int64_t s = 0; s < m->S; s++)
            dd[c * m->S + s] = which < 0 ? LOG_ZERO : state_value(mod, which, (int) s);
}
static void dpm_free(dpm_t *m) {
    for (int64_t i = 0; i <= m->nDiag; i++) free(m->d[i]);
    free(m->d);
}

/* diagonalCalculation :681-712.  `cur` lives in matrix mc; lower/upper on xay-1 of m1, middle on
 * xay-2 of m2 (the expectation pass mixes backward `cur` with forward neighbours :858-862). */
static void diag_calc(const seqs_t *sq, dpm_t *mc, dpm_t *m1, dpm_t *m2, int64_t xay, trans_fn fn,
                      exp_args_t *ea) {
    const int64_t *L = mc->L, *R = mc->R;
    for (int64_t xmy = L[xay]; xmy <= R[xay]; xmy += 2) {
        symbols_t sy;
        get_symbols(sq, diag_x(xay, xmy) - 1, diag_y(xay, xmy) - 1, &sy);
        double *cur = dpm_cell(mc, xay, xmy);
        double *lower = m1 ? dpm_cell(m1, xay - 1, xmy - 1) : NULL;
        double *middle = m2 ? dpm_cell(m2, xay - 2, xmy) : NULL;
        double *upper = m1 ? dpm_cell(m1, xay - 1, xmy + 1) : NULL;
        if (ea) {
            ea->kx = sy.kx; ea->ky = sy.ky; ea->kp = sy.kp;
            ea->ix = diag_x(xay, xmy) - 1; ea->iy = diag_y(xay, xmy) - 1;
        }
        cell_calc(sq->m, cur, lower, middle, upper, &sy, fn, ea);
    }
}

/* cell_dotProduct :391-397 / dpDiagonal_dotProduct :587-597 */
static double diag_dot(const double *a, const double *b, int64_t w, int64_t S) {
    double total = LOG_ZERO;
    for (int64_t c = 0; c < w; c++) {
        double t = a[c * S] + b[c * S];
        for (int64_t s = 1; s < S; s++) t = orc_logAdd(t, a[c * S + s] + b[c * S + s]);
        total = orc_logAdd(total, t);
    }
    return total;
}

/* diagonalCalculationTotalProbability :736-754 */
static double total_probability(const seqs_t *sq, dpm_t *F, dpm_t *B, int64_t xay) {
    int64_t S = F->S;
    double total = diag_dot(F->d[xay], B->d[xay], width_of(F->L, F->R, xay), S);
    double *fd = dpm_diag(F, xay - 1), *bd = dpm_diag(B, xay + 1);
    if (bd != NULL && fd != NULL) {
        int64_t w = width_of(F->L, F->R, xay + 1);
        /* matchDiagonal: shape of backward[xay+1], zeroed, forward step with only `middle` */
        dpm_t tmp = *B; /* shares the band; private pointer table with one diagonal */
        tmp.d = calloc(B->nDiag + 1, sizeof(double *));
        tmp.d[xay + 1] = malloc(sizeof(double) * S * w);
        diag_fill(&tmp, xay + 1, sq->m, -1);
        diag_calc(sq, &tmp, NULL, F, xay + 1, trans_forward, NULL);
        total = orc_logAdd(total, diag_dot(tmp.d[xay + 1], bd, w, S));
        free(tmp.d[xay + 1]);
        free(tmp.d);
    }
    return total;
}

/* diagonalCalculationPosteriorMatchProbs :756-795 */
static void posterior_match_probs(dpm_t *F, dpm_t *B, int64_t xay, double total, double threshold,
                                  orc_result *out) {
    int64_t S = F->S;
    for (int64_t xmy = F->L[xay]; xmy <= F->R[xay]; xmy += 2) {
        int64_t x = diag_x(xay, xmy), y = diag_y(xay, xmy);
        if (x > 0 && y > 0) {
            const double *cf = dpm_cell(F, xay, xmy), *cb = dpm_cell(B, xay, xmy);
            double e = (cf[ST_MATCH] + cb[ST_MATCH]) - total;
            double pp = exp(e);
            if (pp >= threshold) {
                if (pp > 1.0) pp = 1.0;
                pp = floor(pp * ORC_PROB_1);
                result_push(out, (int64_t) pp, x - 1, y - 1, e);
            }
        }
    }
    (void) S;
}

/* diagonalCalculation_Expectations :841-863 */
static void expectations_diag(const seqs_t *sq, dpm_t *F, dpm_t *B, int64_t xay, double total,
                              orc_expectations *hmm) {
    exp_args_t ea = { total, hmm, 0, 0, 0, 0, 0 };
    if (sq->m->kind == ORC_SM3_HDP) {
        ((orc_expectations_h *) hmm)->likelihood += total;
        diag_calc(sq, B, F, F, xay, trans_expect_hdp, &ea);
        return;
    }
    if (sq->m->kind == ORC_SM3_VANILLA) {
        g_vanilla_match = sq->m->match;
        ((orc_expectations_v *) hmm)->likelihood += total;
        diag_calc(sq, B, F, F, xay, trans_expect_vanilla, &ea);
        return;
    }
    if (sq->m->kind == ORC_SM5_SYMBOL) {
        ((orc_expectations5 *) hmm)->likelihood += total;
        diag_calc(sq, B, F, F, xay, trans_expect_sm5, &ea);
        return;
    }
    hmm->likelihood += total;
    diag_calc(sq, B, F, F, xay, trans_expect_sm3, &ea);
}

/* optional dumps for kernel verification */
typedef struct {
    double *F, *B;
    const int64_t *off; /* cell offset of each diagonal */
} dump_t;

/* getPosteriorProbsWithBanding :870-1006 */
static int banded(const seqs_t *sq, const int64_t *anchors, int64_t nAnchors, const orc_params *p,
                  int raggedLeft, int raggedRight, orc_expectations *hmm, orc_result *out,
                  dump_t *dump) {
    int64_t lX = sq->lX, lY = sq->lY, S = sq->m->stateNumber;
    int64_t nDiag = lX + lY;
    if (nDiag == 0) return 0;
    int64_t *L = malloc(sizeof(int64_t) * (nDiag + 1)), *R = malloc(sizeof(int64_t) * (nDiag + 1));
    if (orc_band(anchors, nAnchors, lX, lY, p->diagonalExpansion, L, R) != 0) {
        free(L); free(R);
        return -1;
    }
    int64_t *off = NULL;
    if (dump) {
        off = malloc(sizeof(int64_t) * (nDiag + 2));
        off[0] = 0;
        for (int64_t i = 0; i <= nDiag; i++) off[i + 1] = off[i] + width_of(L, R, i);
        dump->off = off;
    }
    for (int64_t i = 0; i <= nDiag; i++) out->cells += width_of(L, R, i);

    dpm_t F, B;
    dpm_init(&F, nDiag, S, L, R);
    dpm_init(&B, nDiag, S, L, R);
    dpm_create(&F, 0);
    diag_fill(&F, 0, sq->m, raggedLeft ? 1 : 0);
    if (dump && dump->F) memcpy(dump->F, F.d[0], sizeof(double) * S * width_of(L, R, 0));

    int64_t tracedBackTo = 0, totalPosteriorCalculations = 0;
    int64_t cur = 0; /* forward band iterator position (index of last returned diagonal) */
    while (1) {
        cur = cur < nDiag ? cur + 1 : nDiag; /* bandIterator_getNext clamps :213-220 */
        int64_t xay = cur;
        dpm_create(&F, xay);
        diag_fill(&F, xay, sq->m, -1);
        diag_calc(sq, &F, &F, &F, xay, trans_forward, NULL);
        if (dump && dump->F)
            memcpy(dump->F + off[xay] * S, F.d[xay], sizeof(double) * S * width_of(L, R, xay));

        int atEnd = xay == nDiag;
        int tracebackPoint = xay >= tracedBackTo + p->minDiagsBetweenTraceBack
                             && width_of(L, R, xay) <= p->diagonalExpansion * 2 + 1;
        if (atEnd || tracebackPoint) {
            dpm_create(&B, xay);
            diag_fill(&B, xay, sq->m, (atEnd && raggedRight) ? 3 : 2);
            if (xay > tracedBackTo + 1) {
                dpm_create(&B, xay - 1);
                diag_fill(&B, xay - 1, sq->m, -1);
            }
            int64_t d2 = xay;
            int64_t tracedBackFrom = xay - (atEnd ? 0 : p->traceBackDiagonals + 1);
            double totalProbability = LOG_ZERO;
            int64_t calcsThisTraceback = 0;
            while (d2 > tracedBackTo) {
                if (d2 > tracedBackTo + 2) {
                    dpm_create(&B, d2 - 2);
                    diag_fill(&B, d2 - 2, sq->m, -1);
                }
                if (d2 > tracedBackTo + 1) diag_calc(sq, &B, &B, &B, d2, trans_backward, NULL);
                if (d2 <= tracedBackFrom) {
                    if (calcsThisTraceback++ % 10 == 0) {
                        totalProbability = total_probability(sq, &F, &B, d2);
                        result_push_total(out, d2, totalProbability);
                    }
                    if (dump && dump->B)
                        memcpy(dump->B + off[d2] * S, B.d[d2],
                               sizeof(double) * S * width_of(L, R, d2));
                    if (hmm) expectations_diag(sq, &F, &B, d2, totalProbability, hmm);
                    else posterior_match_probs(&F, &B, d2, totalProbability, p->threshold, out);
                    if (d2 < tracedBackFrom || atEnd) dpm_delete(&F, d2);
                }
                if (d2 + 1 <= nDiag) dpm_delete(&B, d2 + 1);
                d2 = d2 > 0 ? d2 - 1 : 0; /* bandIterator_getPrevious */
            }
            tracedBackTo = tracedBackFrom;
            dpm_delete(&B, d2 + 1);
            dpm_delete(&F, d2);
            totalPosteriorCalculations += calcsThisTraceback;
        }
        if (atEnd) break;
    }
    int ok = totalPosteriorCalculations == nDiag ? 0 : -2;
    dpm_free(&F);
    dpm_free(&B);
    free(L); free(R); free(off);
    return ok;
}

static void make_seqs(seqs_t *s, const orc_model *m, const char *x, int64_t lX, const void *y,
                      int64_t lY) {
    s->m = m; s->x = x; s->lX = lX; s->y = y; s->lY = lY;
}

/* slices: sequence_sliceNucleotideSequence2 / sequence_sliceEventSequence2 :287-301 */
static const void *slice_y(const orc_model *m, const void *y, int64_t start) {
    if (m->kind != ORC_SM5_SYMBOL) return ((const double *) y) + 3 * start;
    return ((const char *) y) + start;
}

/* getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps :1356-1422 together with
 * alignedPairCoordinateCorrectionFn :1447-1454 (each sub-list is shifted, then popped from its
 * tail onto the output, i.e. appended reversed). */
int orc_aligned_pairs_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y,
                                    int64_t lY, const int64_t *anchors, int64_t nAnchors,
                                    const orc_params *p, int raggedLeft, int raggedRight,
                                    orc_expectations *hmm, orc_result *out) {
    int64_t cap = nAnchors + 2;
    int64_t *sp = malloc(sizeof(int64_t) * 4 * cap);
    int64_t nSp = orc_split_points(anchors, nAnchors, lX, lY, p->splitMatrixBiggerThanThis,
                                   raggedLeft, raggedRight, sp, cap);
    int64_t j = 0;
    int rc = 0;
    for (int64_t i = 0; i < nSp && rc == 0; i++) {
        int64_t x1 = sp[4 * i], y1 = sp[4 * i + 1], x2 = sp[4 * i + 2], y2 = sp[4 * i + 3];
        int64_t *sub = malloc(sizeof(int64_t) * 2 * (nAnchors + 1));
        int64_t nSub = 0;
        while (j < nAnchors) {
            int64_t ax = anchors[2 * j], ay = anchors[2 * j + 1];
            if (ax + ay >= x2 + y2) break;
            sub[2 * nSub] = ax - x1;
            sub[2 * nSub + 1] = ay - y1;
            nSub++;
            j++;
        }
        seqs_t sq;
        make_seqs(&sq, m, x + x1, x2 - x1, slice_y(m, y, y1), y2 - y1);
        orc_result *part = orc_result_new();
        orc_expectations_h *hh = (hmm && m->kind == ORC_SM3_HDP) ? (orc_expectations_h *) hmm : NULL;
        const int64_t nAssign0 = hh ? hh->n : 0;
        rc = banded(&sq, sub, nSub, p, raggedLeft || i > 0, raggedRight || i < nSp - 1, hmm, part,
                    NULL);
        for (int64_t k = nAssign0; hh && k < hh->n; k++) { /* sub-alignment coordinates -> the read's */
            hh->assign[3 * k + 1] += x1;
            hh->assign[3 * k + 2] += y1;
        }
        for (int64_t k = part->n - 1; k >= 0; k--)
            result_push(out, part->triples[3 * k], part->triples[3 * k + 1] + x1,
                        part->triples[3 * k + 2] + y1, part->logp[k]);
        for (int64_t k = 0; k < part->nTotals; k++)
            result_push_total(out, part->totalsXay[k] + x1 + y1, part->totals[k]);
        out->cells += part->cells;
        orc_result_free(part);
        free(sub);
    }
    free(sp);
    return rc;
}

int orc_banded_dump(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                    const int64_t *anchors, int64_t nAnchors, const orc_params *p, int raggedLeft,
                    int raggedRight, double *dumpF, double *dumpB, orc_result *out) {
    seqs_t sq;
    make_seqs(&sq, m, x, lX, y, lY);
    dump_t d = { dumpF, dumpB, NULL };
    return banded(&sq, anchors, nAnchors, p, raggedLeft, raggedRight, NULL, out, &d);
}

/* getAlignedPairsWithoutBanding :1512-1569 */
int orc_aligned_pairs_without_banding(const orc_model *m, const char *x, int64_t lX, const void *y,
                                      int64_t lY, const orc_params *p, int raggedLeft,
                                      int raggedRight, orc_result *out) {
    seqs_t sq;
    make_seqs(&sq, m, x, lX, y, lY);
    int64_t S = m->stateNumber, nDiag = lX + lY;
    int64_t *L = malloc(sizeof(int64_t) * (nDiag + 1)), *R = malloc(sizeof(int64_t) * (nDiag + 1));
    if (orc_band(NULL, 0, lX, lY, 2, L, R) != 0) { free(L); free(R); return -1; }
    dpm_t F, B;
    dpm_init(&F, nDiag, S, L, R);
    dpm_init(&B, nDiag, S, L, R);
    for (int64_t i = 0; i <= nDiag; i++) {
        dpm_create(&B, i); diag_fill(&B, i, m, -1);
        dpm_create(&F, i); diag_fill(&F, i, m, -1);
        out->cells += width_of(L, R, i);
    }
    diag_fill(&F, 0, m, raggedLeft ? 1 : 0);
    diag_fill(&B, nDiag, m, raggedRight ? 3 : 2);
    for (int64_t i = 0; i <= nDiag; i++) diag_calc(&sq, &F, &F, &F, i, trans_forward, NULL);
    for (int64_t i = nDiag; i > 0; i--) diag_calc(&sq, &B, &B, &B, i, trans_backward, NULL);
    double total = total_probability(&sq, &F, &B, nDiag);
    result_push_total(out, nDiag, total);
    for (int64_t i = 0; i <= nDiag; i++) posterior_match_probs(&F, &B, i, total, p->threshold, out);
    dpm_free(&F);
    dpm_free(&B);
    free(L); free(R);
    return 0;
}

orc_expectations_h *orc_expectations_h_new(double threshold) {
    orc_expectations_h *h = calloc(1, sizeof *h);
    h->threshold = threshold;
    return h;
}
void orc_expectations_h_free(orc_expectations_h *h) {
    free(h->assign);
    free(h->logp);
    free(h);
}
int orc_expectations_h_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                                     const int64_t *anchors, int64_t nAnchors, const orc_params *p,
                                     int raggedLeft, int raggedRight, orc_expectations_h *hmm) {
    if (m->kind != ORC_SM3_HDP) return -2;
    orc_result *scratch = orc_result_new();
    int rc = orc_aligned_pairs_using_anchors(m, x, lX, y, lY, anchors, nAnchors, p, raggedLeft, raggedRight,
                                             (orc_expectations *) hmm, scratch);
    orc_result_free(scratch);
    return rc;
}

int orc_expectations_v_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                                     const int64_t *anchors, int64_t nAnchors, const orc_params *p,
                                     int raggedLeft, int raggedRight, orc_expectations_v *hmm) {
    if (m->kind != ORC_SM3_VANILLA) return -2;
    orc_result *scratch = orc_result_new();
    int rc = orc_aligned_pairs_using_anchors(m, x, lX, y, lY, anchors, nAnchors, p, raggedLeft, raggedRight,
                                             (orc_expectations *) hmm, scratch);
    orc_result_free(scratch);
    return rc;
}

int orc_expectations5_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                                    const int64_t *anchors, int64_t nAnchors, const orc_params *p,
                                    int raggedLeft, int raggedRight, orc_expectations5 *hmm) {
    if (m->kind != ORC_SM5_SYMBOL) return -2;
    orc_result *scratch = orc_result_new();
    int rc = orc_aligned_pairs_using_anchors(m, x, lX, y, lY, anchors, nAnchors, p, raggedLeft, raggedRight,
                                             (orc_expectations *) hmm, scratch);
    orc_result_free(scratch);
    return rc;
}

void orc_expectations5_normalize(orc_expectations5 *e) {
    for (int from = 0; from < 5; from++) {
        double total = 0.0;
        for (int to = 0; to < 5; to++) total += e->transitions[from * 5 + to];
        for (int to = 0; to < 5; to++) e->transitions[from * 5 + to] = e->transitions[from * 5 + to] / total;
    }
    for (int s = 0; s < 5; s++) {
        double total = 0.0;
        for (int i = 0; i < 16; i++) total += e->emissions[s * 16 + i];
        for (int i = 0; i < 16; i++) e->emissions[s * 16 + i] = e->emissions[s * 16 + i] / total;
    }
}

/* continuousPairHmm_normalize impl/continuousHmm.c:174-191 + hmmDiscrete_normalize2 :125-136 */
void orc_expectations_normalize(orc_expectations *e) {
    for (int from = 0; from < 3; from++) {
        double total = 0.0;
        for (int to = 0; to < 3; to++) total += e->transitions[from * 3 + to];
        for (int to = 0; to < 3; to++) e->transitions[from * 3 + to] = e->transitions[from * 3 + to] / total;
    }
    double total = 0.0;
    for (int i = 0; i < ORC_NUM_KMERS; i++) total += e->kmerGap[i];
    for (int i = 0; i < ORC_NUM_KMERS; i++) e->kmerGap[i] = e->kmerGap[i] / total;
}
