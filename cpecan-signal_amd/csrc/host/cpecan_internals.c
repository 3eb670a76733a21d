/*
 * cpecan_internals.c -- the plain host C behind the function-pointer members of the reference's StateMachine and Hmm
 * "classes", and the DP internals the reference exports and tests (inc/pairwiseAligner.h:190-267):
 *
 *   - the emission plug-ins a caller hands to the stateMachine*_construct functions (impl/stateMachine.c:175-630),
 *   - startStateProb / endStateProb / ragged*StateProb / cellCalculate of the four machines on the path
 *     (impl/stateMachine.c:741-895, 1166-1400),
 *   - cell_calculateForward / Backward, cell_dotProduct, DpDiagonal, DpMatrix, diagonalCalculation*
 *     (impl/pairwiseAligner.c:357-866),
 *   - the Hmm subclasses of the signal machines (impl/continuousHmm.c).
 *
 * None of this is on the aligner's hot path: getAlignedPairsUsingAnchors / getExpectationsUsingAnchors recognise the
 * known machines and run the gfx950 kernels (cpecan_api.c).  These functions exist so that a caller or a test that
 * drives the recurrence cell by cell, or diagonal by diagonal, through the reference's own interface finds it there,
 * with the reference's arithmetic (same logAdd, same order of transitions).
 */
#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cpecan_host_private.h"

#define die cpecan_die

/* ------------------------------------------------------------------------------------------------ */
/* emission plug-ins                                                                                */
/* ------------------------------------------------------------------------------------------------ */
/* entry j of k-mer k in a [correlation | 5 per k-mer] table; an index past the table (a k-mer holding a character
 * other than ACGT) reads as 0.0 (impl/stateMachine.c:220-240) */
static double model_entry(const double *table, int64_t k, int j) {
    return k > NUM_OF_KMERS ? 0.0 : table[1 + k * MODEL_PARAMS + j];
}
static double log_normal(double x, double mu, double sigma) { /* :333-343 */
    if (sigma == 0.0) return LOG_ZERO;
    const double a = (x - mu) / sigma;
    return -0.91893853320467267 - log(sigma) + (-0.5 * a * a);
}
static double log_inverse_gaussian(double x, double mu, double lambda) { /* :322-331 */
    const double a = (x - mu) / mu;
    return (log(lambda) - 1.8378770664093453 - 3 * log(x) - lambda * a * a / x) / 2;
}

double emissions_kmer_getGapProb(const double *emissionGapProbs, void *kmer) {
    const int64_t i = emissions_discrete_getKmerIndex(kmer);
    return i > NUM_OF_KMERS ? LOG_ZERO : emissionGapProbs[i];
}
double emissions_signal_strawManGetKmerEventMatchProb(const double *eventModel, void *kmer, void *event) {
    const double *e = event;
    const int64_t k = emissions_discrete_getKmerIndex(kmer);
    const double level = log_normal(e[0], model_entry(eventModel, k, 0), model_entry(eventModel, k, 1));
    const double noise = log_normal(e[1], model_entry(eventModel, k, 2), model_entry(eventModel, k, 3));
    return level + noise;
}
/* the x element of the vanilla machine is sequence_getKmer2's pair "previous k-mer, this k-mer": 7 characters */
double emissions_signal_getEventMatchProbWithTwoDists(const double *eventModel, void *kmers, void *event) {
    const double *e = event;
    const int64_t k = emissions_discrete_getKmerIndex((char *) kmers + 1);
    const double level = log_normal(e[0], model_entry(eventModel, k, 0), model_entry(eventModel, k, 1));
    const double noise = log_inverse_gaussian(e[1], model_entry(eventModel, k, 2), model_entry(eventModel, k, 4));
    return level + noise;
}
int64_t emissions_signal_getKmerSkipBin(double *matchModel, void *kmers);
/* ---- small emission helpers the reference exports beside the ones its machines use (impl/stateMachine.c,
 * impl/emissionMatrix.c): host functions for callers that assemble machines of their own from them ---- */
/* the 25 x 25 match table of two-base "k-mers" over ACGTN (impl/emissionMatrix.c:11-53): the sum of the two bases'
 * log-probabilities -- identical, transition (A<->G, C<->T), transversion, or N */
void emissions_kmer_setMatchProbsToDefaults(double *emissionMatchProbs) {
    const double M = -2.1149196655034745, V = -4.5691014376830479, S = -3.9833860032220842, N = -2.772588722;
    for (int a = 0; a < 25; a++)
        for (int b = 0; b < 25; b++) {
            double e[2];
            for (int pos = 0; pos < 2; pos++) {
                const int x = pos == 0 ? a / 5 : a % 5, y = pos == 0 ? b / 5 : b % 5;
                e[pos] = x == 4 || y == 4 ? N : x == y ? M : (x ^ y) == 2 ? S : V;
            }
            emissionMatchProbs[a * 25 + b] = e[0] + e[1];
        }
}
void emissions_kmer_setGapProbsToDefaults(double *emissionGapProbs) { /* :55-69: log(0.2) + log(0.2) */
    for (int i = 0; i < 25; i++) emissionGapProbs[i] = -3.2188758248682006;
}
void emissions_discrete_initEmissionsToZero(StateMachine *sM) { /* :94-102 */
    const size_t n = (size_t) sM->parameterSetSize;
    sM->EMISSION_GAP_X_PROBS = calloc(n, sizeof(double));
    sM->EMISSION_GAP_Y_PROBS = calloc(n, sizeof(double));
    sM->EMISSION_MATCH_PROBS = calloc(n * n, sizeof(double));
}
double emissions_kmer_getMatchProb(const double *emissionMatchProbs, void *x, void *y) { /* :189-194 */
    return emissionMatchProbs[emissions_discrete_getKmerIndex(x) * NUM_OF_KMERS + emissions_discrete_getKmerIndex(y)];
}
double emissions_signal_logGaussMatchProb(const double *eventModel, void *kmer, void *event) { /* :473-497 */
    const int64_t k = emissions_discrete_getKmerIndex((char *) kmer + 1);
    const double sd = model_entry(eventModel, k, 1), a = (*(double *) event - model_entry(eventModel, k, 0)) / sd;
    return log(0.3989422804014327) - log(sd) + (-0.5 * a * a);
}
double emissions_signal_getBivariateGaussPdfMatchProb(const double *eventModel, void *kmer, void *event) { /* :556-593 */
    const double *e = event;
    const double rho = eventModel[0], rhoSq = rho * rho;
    const int64_t k = emissions_discrete_getKmerIndex((char *) kmer + 1);
    const double levelMean = model_entry(eventModel, k, 0), levelSd = model_entry(eventModel, k, 1);
    const double noiseMean = model_entry(eventModel, k, 2), noiseSd = model_entry(eventModel, k, 3);
    const double expC = -1 / (2 * (1 - rhoSq));
    const double xu = (e[0] - levelMean) / levelSd, yu = (e[1] - noiseMean) / noiseSd;
    const double a = expC * ((xu * xu) + (yu * yu) - (2 * rho * xu * yu));
    return (-1.8378770664093453 - log(levelSd * noiseSd * sqrt(1 - rhoSq))) + a;
}
/* the posterior of n k-mers given the event's duration under a Poisson model (:345-370, :551-554) */
double emissions_signal_getDurationProb(void *event, int64_t n) {
    static const double logFactorial[6] = { 0.0, 0.0, 0.69314718056, 1.79175946923, 3.17805383035, 4.78749174278 };
    if (n < 0 || n > 5) die("emissions_signal_getDurationProb: n = %lld (at most 5)", (long long) n);
    const double lambda = ((double *) event)[2] / 0.00332005312085;
    return (n + 1) * 0.1397619423751586 + n * log(lambda) - logFactorial[n] - 2 * lambda;
}
/* the skip probability of the bin the two k-mers' level difference falls in -- NOT in log space (:429-471) */
double emissions_signal_getKmerSkipProb(StateMachine *sM, void *kmers) {
    return sM->EMISSION_GAP_X_PROBS[emissions_signal_getKmerSkipBin(sM->EMISSION_MATCH_PROBS, kmers)];
}
void emissions_signal_scaleModelNoiseOnly(StateMachine *sM, double scale, double shift, double var, double scale_sd,
                                          double var_sd) { /* :653-672: emissions_signal_scaleModel without the level mean */
    (void) scale; (void) shift;
    double *m = sM->EMISSION_MATCH_PROBS;
    for (int64_t i = 1; i < (sM->parameterSetSize * MODEL_PARAMS) + 1; i += MODEL_PARAMS) {
        m[i + 1] = m[i + 1] * var;
        m[i + 2] = m[i + 2] * scale_sd;
        m[i + 4] = m[i + 4] * var_sd;
        m[i + 3] = sqrt(pow(m[i + 2], 3.0) / m[i + 4]);
    }
}
void stateMachine3_setTransitionsToNucleotideDefaults(StateMachine *sM) { /* :1265-1276 */
    StateMachine3 *s = (StateMachine3 *) sM;
    s->TRANSITION_MATCH_CONTINUE = -0.030064059121770816;
    s->TRANSITION_MATCH_FROM_GAP_X = s->TRANSITION_MATCH_FROM_GAP_Y = -1.272871422049609;
    s->TRANSITION_GAP_OPEN_X = s->TRANSITION_GAP_OPEN_Y = -4.21256642;
    s->TRANSITION_GAP_EXTEND_X = s->TRANSITION_GAP_EXTEND_Y = -0.3388262689231553;
    s->TRANSITION_GAP_SWITCH_TO_X = s->TRANSITION_GAP_SWITCH_TO_Y = -4.910694825551255;
}
char *diagonal_getString(Diagonal diagonal) { /* impl/pairwiseAligner.c:81-84; the caller frees */
    char *out = malloc(128);
    snprintf(out, 128, "Diagonal, xay: %lld xmyL %lld, xmyR: %lld", (long long) diagonal.xay, (long long) diagonal.xmyL,
             (long long) diagonal.xmyR);
    return out;
}
int sortByXPlusYCoordinate(const void *i, const void *j) { /* :1011-1015: (x, y) pairs */
    const int64_t k = stIntTuple_get((stIntTuple *) i, 0) + stIntTuple_get((stIntTuple *) i, 1);
    const int64_t l = stIntTuple_get((stIntTuple *) j, 0) + stIntTuple_get((stIntTuple *) j, 1);
    return k > l ? 1 : (k < l ? -1 : 0);
}
int sortByXPlusYCoordinate2(const void *i, const void *j) { /* :1019-1023: (score, x, y) triples */
    const int64_t k = stIntTuple_get((stIntTuple *) i, 1) + stIntTuple_get((stIntTuple *) i, 2);
    const int64_t l = stIntTuple_get((stIntTuple *) j, 1) + stIntTuple_get((stIntTuple *) j, 2);
    return k > l ? 1 : (k < l ? -1 : 0);
}

int64_t emissions_signal_getKmerSkipBin(double *matchModel, void *kmers) {
    const int64_t before = emissions_discrete_getKmerIndex(kmers);
    const int64_t here = emissions_discrete_getKmerIndex((char *) kmers + 1);
    const double d = fabs(model_entry(matchModel, here, 0) - model_entry(matchModel, before, 0));
    const int64_t bin = (int64_t) (d / 0.5);
    return bin >= 30 ? 29 : bin;
}
double emissions_signal_getBetaOrAlphaSkipProb(StateMachine *sM, void *kmers, bool getAlpha) {
    const int64_t bin = emissions_signal_getKmerSkipBin(sM->EMISSION_MATCH_PROBS, kmers);
    return sM->EMISSION_GAP_X_PROBS[getAlpha ? bin + 30 : bin];
}
void emissions_signal_initEmissionsToZero(StateMachine *sM, int64_t nbSkipParams) {
    const size_t table = 1 + (size_t) sM->parameterSetSize * MODEL_PARAMS;
    sM->EMISSION_GAP_X_PROBS = calloc((size_t) nbSkipParams, sizeof(double));
    sM->EMISSION_GAP_Y_PROBS = calloc(table, sizeof(double));
    sM->EMISSION_MATCH_PROBS = calloc(table, sizeof(double));
}

/* get_nanopore_kmer_density (impl/nanopore_hdp.c:390) -> dir_proc_density (impl/hdp.c:2577-2601): the spline of the
 * nearest observed Dirichlet process on the sampling grid (grid_spline_interp, impl/hdp_math_utils.c:471-495),
 * clamped at zero.  A linear density, used by the HDP machine where a log-probability belongs (quirk Q1). */
double get_nanopore_kmer_density(NanoporeHDP *h, void *kmer, void *x) {
    const char *k = kmer;
    const double q = *(double *) x;
    int64_t id = 0;
    for (int i = 0; i < h->kmerLength; i++) {
        int64_t j = 0;
        while (j < h->alphabetSize && k[i] != h->alphabet[j]) j++;
        if (j == h->alphabetSize)
            die("vanillaAlign - ERROR: K-mer contains character outside alphabet. Got offending kmer is: %.*s. "
                "alphabet is %s", (int) h->kmerLength, k, h->alphabet);
        id = id * h->alphabetSize + j;
    }
    const int64_t row = h->kmerRow[id], n = h->gridLength - 1;
    const double *gx = h->grid, *y = h->y + row * h->gridLength, *s = h->slope + row * h->gridLength;
    double r;
    if (q <= gx[0]) r = y[0] - s[0] * (gx[0] - q);
    else if (q >= gx[n]) r = y[n] + s[n] * (q - gx[n]);
    else {
        const double dx = gx[1] - gx[0];
        const int64_t il = (int64_t) ((q - gx[0]) / dx), ir = il + 1;
        const double dy = y[ir] - y[il];
        const double a = s[il] * dx - dy, b = dy - s[ir] * dx;
        const double tl = (q - gx[il]) / dx, tr = 1.0 - tl;
        r = tr * y[il] + tl * y[ir] + tl * tr * (a * tr + b * tl);
    }
    return r > 0.0 ? r : 0.0;
}

/* ------------------------------------------------------------------------------------------------ */
/* the machines' member functions                                                                   */
/* ------------------------------------------------------------------------------------------------ */
static void state_check(StateMachine *sM, int64_t state) {
    if (state < 0 || state >= sM->stateNumber) die("cpecan: state %lld of a %lld-state machine", (long long) state,
                                                    (long long) sM->stateNumber);
}
static double only_match_starts(StateMachine *sM, int64_t state) { /* :741, :1166: the 3- and 5-state start vector */
    state_check(sM, state);
    return state == match ? 0 : LOG_ZERO;
}
static double sm3_ragged_start(StateMachine *sM, int64_t state) {
    state_check(sM, state);
    return state == shortGapX || state == shortGapY ? 0 : LOG_ZERO;
}
static double sm3_end(StateMachine *sM, int64_t state) {
    const StateMachine3 *s = (StateMachine3 *) sM;
    state_check(sM, state);
    const double v[3] = { s->TRANSITION_MATCH_CONTINUE, s->TRANSITION_MATCH_FROM_GAP_X, s->TRANSITION_MATCH_FROM_GAP_Y };
    return v[state];
}
static double sm3_ragged_end(StateMachine *sM, int64_t state) {
    const StateMachine3 *s = (StateMachine3 *) sM;
    state_check(sM, state);
    const double v[3] = { (s->TRANSITION_GAP_OPEN_X + s->TRANSITION_GAP_OPEN_Y) / 2.0, s->TRANSITION_GAP_EXTEND_X,
                          s->TRANSITION_GAP_EXTEND_Y };
    return v[state];
}
static double vanilla_end(StateMachine *sM, int64_t state) {
    const StateMachine3Vanilla *s = (StateMachine3Vanilla *) sM;
    state_check(sM, state);
    const double v[3] = { s->DEFAULT_END_MATCH_PROB, s->DEFAULT_END_FROM_X_PROB, s->DEFAULT_END_FROM_Y_PROB };
    return v[state];
}
static double vanilla_ragged_end(StateMachine *sM, int64_t state) {
    const StateMachine3Vanilla *s = (StateMachine3Vanilla *) sM;
    state_check(sM, state);
    const double v[3] = { (s->DEFAULT_END_FROM_X_PROB + s->DEFAULT_END_FROM_Y_PROB) / 2.0, s->DEFAULT_END_FROM_X_PROB,
                          s->DEFAULT_END_FROM_Y_PROB };
    return v[state];
}
static double sm5_ragged_start(StateMachine *sM, int64_t state) {
    state_check(sM, state);
    return state == longGapX || state == longGapY ? 0 : LOG_ZERO;
}
static double sm5_end(StateMachine *sM, int64_t state) {
    const StateMachine5 *s = (StateMachine5 *) sM;
    state_check(sM, state);
    const double v[5] = { s->TRANSITION_MATCH_CONTINUE, s->TRANSITION_MATCH_FROM_SHORT_GAP_X,
                          s->TRANSITION_MATCH_FROM_SHORT_GAP_Y, s->TRANSITION_MATCH_FROM_LONG_GAP_X,
                          s->TRANSITION_MATCH_FROM_LONG_GAP_Y };
    return v[state];
}
static double sm5_ragged_end(StateMachine *sM, int64_t state) {
    const StateMachine5 *s = (StateMachine5 *) sM;
    state_check(sM, state);
    const double v[5] = { s->TRANSITION_GAP_LONG_OPEN_X, s->TRANSITION_GAP_LONG_OPEN_X, s->TRANSITION_GAP_LONG_OPEN_Y,
                          s->TRANSITION_GAP_LONG_EXTEND_X, s->TRANSITION_GAP_LONG_EXTEND_Y };
    return v[state];
}

/* One arc of a machine: from-state, to-state, transition log-probability.  A cell is computed by walking the arcs of
 * its lower (x - 1), middle (x - 1, y - 1) and upper (y - 1) neighbour in the reference's order; the order is part of
 * the result because logAdd is approximate. */
typedef struct {
    int64_t from, to;
    double tP;
} Arc;
static void walk(const Arc *arcs, int n, double *neighbour, double *current, double eP, DoTransitionFn doTransition,
                 void *extraArgs) {
    for (int i = 0; i < n; i++) doTransition(neighbour, current, arcs[i].from, arcs[i].to, eP, arcs[i].tP, extraArgs);
}

/* stateMachine3_cellCalculate :1305-1336 */
static void sm3_cell(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                     void *cY, DoTransitionFn doTransition, void *extraArgs) {
    const StateMachine3 *s = (StateMachine3 *) sM;
    if (lower) {
        const Arc a[3] = { { match, shortGapX, s->TRANSITION_GAP_OPEN_X },
                           { shortGapX, shortGapX, s->TRANSITION_GAP_EXTEND_X },
                           { shortGapY, shortGapX, s->TRANSITION_GAP_SWITCH_TO_X } };
        walk(a, 3, lower, current, s->getXGapProbFcn(sM->EMISSION_GAP_X_PROBS, cX), doTransition, extraArgs);
    }
    if (middle) {
        const Arc a[3] = { { match, match, s->TRANSITION_MATCH_CONTINUE },
                           { shortGapX, match, s->TRANSITION_MATCH_FROM_GAP_X },
                           { shortGapY, match, s->TRANSITION_MATCH_FROM_GAP_Y } };
        walk(a, 3, middle, current, s->getMatchProbFcn(sM->EMISSION_MATCH_PROBS, cX, cY), doTransition, extraArgs);
    }
    if (upper) { /* no gapX -> gapY arc: a skipped k-mer is not followed by an extra event */
        const Arc a[2] = { { match, shortGapY, s->TRANSITION_GAP_OPEN_Y },
                           { shortGapY, shortGapY, s->TRANSITION_GAP_EXTEND_Y } };
        walk(a, 2, upper, current, s->getYGapProbFcn(sM->EMISSION_GAP_Y_PROBS, cX, cY), doTransition, extraArgs);
    }
}
/* stateMachine4_cellCalculate :867-897 and its state vectors :791-829 (start: stateMachine5_startStateProb :743) */
static void sm4_cell(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                     void *cY, DoTransitionFn doTransition, void *extraArgs) {
    const StateMachine4 *s = (StateMachine4 *) sM;
    if (lower) {
        const Arc a[5] = { { match, shortGapX, s->TRANSITION_GAP_SHORT_OPEN_X },
                           { shortGapX, shortGapX, s->TRANSITION_GAP_SHORT_EXTEND_X },
                           { match, longGapX, s->TRANSITION_GAP_LONG_OPEN_X },
                           { longGapX, longGapX, s->TRANSITION_GAP_LONG_EXTEND_X },
                           { shortGapY, longGapX, s->TRANSITION_GAP_LONG_SWITCH_TO_X } };
        walk(a, 5, lower, current, s->getXGapProbFcn(sM->EMISSION_GAP_X_PROBS, cX), doTransition, extraArgs);
    }
    if (middle) {
        const Arc a[4] = { { match, match, s->TRANSITION_MATCH_CONTINUE },
                           { shortGapX, match, s->TRANSITION_MATCH_FROM_SHORT_GAP_X },
                           { shortGapY, match, s->TRANSITION_MATCH_FROM_SHORT_GAP_Y },
                           { longGapX, match, s->TRANSITION_MATCH_FROM_LONG_GAP_X } };
        walk(a, 4, middle, current, s->getMatchProbFcn(sM->EMISSION_MATCH_PROBS, cX, cY), doTransition, extraArgs);
    }
    if (upper) {
        const Arc a[2] = { { match, shortGapY, s->TRANSITION_GAP_SHORT_OPEN_Y },
                           { shortGapY, shortGapY, s->TRANSITION_GAP_SHORT_EXTEND_Y } };
        walk(a, 2, upper, current, s->getYGapProbFcn(sM->EMISSION_GAP_Y_PROBS, cX, cY), doTransition, extraArgs);
    }
}
static double sm4_ragged_start(StateMachine *sM, int64_t state) {
    state_check(sM, state);
    return (state == longGapX || state == shortGapY) ? 0 : LOG_ZERO;
}
static double sm4_end(StateMachine *sM, int64_t state) {
    const StateMachine4 *s = (StateMachine4 *) sM;
    state_check(sM, state);
    const double v[4] = { s->TRANSITION_MATCH_CONTINUE, s->TRANSITION_MATCH_FROM_SHORT_GAP_X,
                          s->TRANSITION_MATCH_FROM_SHORT_GAP_Y, s->TRANSITION_MATCH_FROM_LONG_GAP_X };
    return v[state];
}
static double sm4_ragged_end(StateMachine *sM, int64_t state) {
    const StateMachine4 *s = (StateMachine4 *) sM;
    state_check(sM, state);
    return state == longGapX ? s->TRANSITION_GAP_LONG_EXTEND_X : s->TRANSITION_GAP_LONG_OPEN_X;
}
void cpecan_sm4_set_functions(StateMachine4 *s) {
    s->model.startStateProb = only_match_starts;
    s->model.raggedStartStateProb = sm4_ragged_start;
    s->model.endStateProb = sm4_end;
    s->model.raggedEndStateProb = sm4_ragged_end;
    s->model.cellCalculate = sm4_cell;
}
/* stateMachine4_construct :960-1037: the transitions are "from a template read" */
StateMachine *stateMachine4_construct(StateMachineType type, int64_t parameterSetSize,
                                      void (*setEmissionsToDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                      double (*gapXProbFcn)(const double *, void *),
                                      double (*gapYProbFcn)(const double *, void *, void *),
                                      double (*matchProbFcn)(const double *, void *, void *),
                                      void (*cellCalcUpdateFcn)(double *, double *, int64_t, int64_t, double, double,
                                                                void *)) {
    if (type != fourState) die("Tried to make four-state stateMachine, with wrong type");
    StateMachine4 *s = calloc(1, sizeof *s);
    s->model.type = type;
    s->model.parameterSetSize = parameterSetSize;
    s->model.stateNumber = 4;
    s->model.matchState = match;
    cpecan_sm4_set_functions(s);
    s->model.cellCalculateUpdateExpectations = cellCalcUpdateFcn;
    s->getXGapProbFcn = gapXProbFcn;
    s->getYGapProbFcn = gapYProbFcn;
    s->getMatchProbFcn = matchProbFcn;
    s->TRANSITION_MATCH_CONTINUE = -0.23552123624314988;
    s->TRANSITION_GAP_SHORT_OPEN_X = -1.6269694202638481;
    s->TRANSITION_GAP_SHORT_OPEN_Y = -4.7241893208381773;
    s->TRANSITION_GAP_LONG_OPEN_X = -5.4173365013981227;
    s->TRANSITION_GAP_SHORT_EXTEND_X = -1.6269694202638481;
    s->TRANSITION_MATCH_FROM_SHORT_GAP_X = -0.21880828092192281;
    s->TRANSITION_GAP_LONG_EXTEND_X = -0.003442492794189331;
    s->TRANSITION_MATCH_FROM_LONG_GAP_X = -5.6732801731704612;
    s->TRANSITION_MATCH_FROM_SHORT_GAP_Y = -0.013406326748077823;
    s->TRANSITION_GAP_SHORT_EXTEND_Y = -4.724189320832104;
    s->TRANSITION_GAP_LONG_SWITCH_TO_X = -5.4173365013920494;
    setEmissionsToDefaults((StateMachine *) s, parameterSetSize);
    return (StateMachine *) s;
}

/* stateMachine3HDP_cellCalculate :1338-1370: as above with the densities of the HDP and a flat log(0.1) k-mer gap */
static void sm3hdp_cell(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                        void *cY, DoTransitionFn doTransition, void *extraArgs) {
    const StateMachine3_HDP *s = (StateMachine3_HDP *) sM;
    if (lower) {
        const Arc a[3] = { { match, shortGapX, s->TRANSITION_GAP_OPEN_X },
                           { shortGapX, shortGapX, s->TRANSITION_GAP_EXTEND_X },
                           { shortGapY, shortGapX, s->TRANSITION_GAP_SWITCH_TO_X } };
        walk(a, 3, lower, current, -2.3025850929940455, doTransition, extraArgs);
    }
    if (middle) {
        const Arc a[3] = { { match, match, s->TRANSITION_MATCH_CONTINUE },
                           { shortGapX, match, s->TRANSITION_MATCH_FROM_GAP_X },
                           { shortGapY, match, s->TRANSITION_MATCH_FROM_GAP_Y } };
        walk(a, 3, middle, current, s->getMatchProbFcn(s->hdpModel, cX, cY), doTransition, extraArgs);
    }
    if (upper) {
        const Arc a[2] = { { match, shortGapY, s->TRANSITION_GAP_OPEN_Y },
                           { shortGapY, shortGapY, s->TRANSITION_GAP_EXTEND_Y } };
        walk(a, 2, upper, current, s->getYGapProbFcn(s->hdpModel, cX, cY), doTransition, extraArgs);
    }
}
/* stateMachine3Vanilla_cellCalculate :1372-1412: transitions from the skip bin of the k-mer pair; the float
 * literals (1.0f, the 0.55f of TRANSITION_E_TO_E) are the reference's */
static void vanilla_cell(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                         void *cY, DoTransitionFn doTransition, void *extraArgs) {
    const StateMachine3Vanilla *s = (StateMachine3Vanilla *) sM;
    const double a_mx = s->getKmerSkipProb(sM, cX, 0);
    const double a_my = (1 - a_mx) * s->TRANSITION_M_TO_Y_NOT_X;
    const double a_mm = 1.0f - a_my - a_mx;
    const double a_yy = s->TRANSITION_E_TO_E;
    const double a_ym = 1.0f - a_yy;
    const double a_xx = s->getKmerSkipProb(sM, cX, 1);
    const double a_xm = 1.0f - a_xx;
    if (lower) {
        const Arc a[2] = { { match, shortGapX, log(a_mx) }, { shortGapX, shortGapX, log(a_xx) } };
        walk(a, 2, lower, current, 0, doTransition, extraArgs);
    }
    if (middle) {
        const Arc a[3] = { { match, match, log(a_mm) }, { shortGapX, match, log(a_xm) }, { shortGapY, match, log(a_ym) } };
        walk(a, 3, middle, current, s->getMatchProbFcn(sM->EMISSION_MATCH_PROBS, cX, cY), doTransition, extraArgs);
    }
    if (upper) {
        const Arc a[2] = { { match, shortGapY, log(a_my) }, { shortGapY, shortGapY, log(a_yy) } };
        walk(a, 2, upper, current, s->getScaledMatchProbFcn(sM->EMISSION_GAP_Y_PROBS, cX, cY), doTransition, extraArgs);
    }
}
/* stateMachine5_cellCalculate :836-867 */
static void sm5_cell(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                     void *cY, DoTransitionFn doTransition, void *extraArgs) {
    const StateMachine5 *s = (StateMachine5 *) sM;
    if (lower) {
        const Arc a[4] = { { match, shortGapX, s->TRANSITION_GAP_SHORT_OPEN_X },
                           { shortGapX, shortGapX, s->TRANSITION_GAP_SHORT_EXTEND_X },
                           { match, longGapX, s->TRANSITION_GAP_LONG_OPEN_X },
                           { longGapX, longGapX, s->TRANSITION_GAP_LONG_EXTEND_X } };
        walk(a, 4, lower, current, s->getXGapProbFcn(sM->EMISSION_GAP_X_PROBS, cX), doTransition, extraArgs);
    }
    if (middle) {
        const Arc a[5] = { { match, match, s->TRANSITION_MATCH_CONTINUE },
                           { shortGapX, match, s->TRANSITION_MATCH_FROM_SHORT_GAP_X },
                           { shortGapY, match, s->TRANSITION_MATCH_FROM_SHORT_GAP_Y },
                           { longGapX, match, s->TRANSITION_MATCH_FROM_LONG_GAP_X },
                           { longGapY, match, s->TRANSITION_MATCH_FROM_LONG_GAP_Y } };
        walk(a, 5, middle, current, s->getMatchProbFcn(sM->EMISSION_MATCH_PROBS, cX, cY), doTransition, extraArgs);
    }
    if (upper) {
        const Arc a[4] = { { match, shortGapY, s->TRANSITION_GAP_SHORT_OPEN_Y },
                           { shortGapY, shortGapY, s->TRANSITION_GAP_SHORT_EXTEND_Y },
                           { match, longGapY, s->TRANSITION_GAP_LONG_OPEN_Y },
                           { longGapY, longGapY, s->TRANSITION_GAP_LONG_EXTEND_Y } };
        walk(a, 4, upper, current, s->getYGapProbFcn(sM->EMISSION_GAP_Y_PROBS, cY), doTransition, extraArgs);
    }
}

void cpecan_sm3_set_functions(StateMachine3 *s) {
    s->model.startStateProb = only_match_starts;
    s->model.raggedStartStateProb = sm3_ragged_start;
    s->model.endStateProb = sm3_end;
    s->model.raggedEndStateProb = sm3_ragged_end;
    s->model.cellCalculate = sm3_cell;
}
void cpecan_sm3hdp_set_functions(StateMachine3_HDP *s) {
    /* StateMachine3_HDP begins with the nine transitions of StateMachine3, so its state vectors are sm3's */
    s->model.startStateProb = only_match_starts;
    s->model.raggedStartStateProb = sm3_ragged_start;
    s->model.endStateProb = sm3_end;
    s->model.raggedEndStateProb = sm3_ragged_end;
    s->model.cellCalculate = sm3hdp_cell;
}
void cpecan_sm3vanilla_set_functions(StateMachine3Vanilla *s) {
    s->model.startStateProb = only_match_starts;
    s->model.raggedStartStateProb = sm3_ragged_start;
    s->model.endStateProb = vanilla_end;
    s->model.raggedEndStateProb = vanilla_ragged_end;
    s->model.cellCalculate = vanilla_cell;
}
void cpecan_sm5_set_functions(StateMachine5 *s) {
    s->model.startStateProb = only_match_starts;
    s->model.raggedStartStateProb = sm5_ragged_start;
    s->model.endStateProb = sm5_end;
    s->model.raggedEndStateProb = sm5_ragged_end;
    s->model.cellCalculate = sm5_cell;
}
int cpecan_sm_functions_known(StateMachine *sM) {
    const void *cell = (const void *) sM->cellCalculate;
    if (cell == (const void *) sm5_cell) {
        const StateMachine5 *s = (StateMachine5 *) sM;
        return s->getXGapProbFcn == emissions_symbol_getGapProb && s->getYGapProbFcn == emissions_symbol_getGapProb &&
               s->getMatchProbFcn == emissions_symbol_getMatchProb;
    }
    if (cell == (const void *) sm3_cell) {
        const StateMachine3 *s = (StateMachine3 *) sM;
        return s->getXGapProbFcn == emissions_kmer_getGapProb &&
               s->getYGapProbFcn == emissions_signal_strawManGetKmerEventMatchProb &&
               s->getMatchProbFcn == emissions_signal_strawManGetKmerEventMatchProb;
    }
    if (cell == (const void *) sm4_cell) {
        const StateMachine4 *s = (StateMachine4 *) sM;
        return s->getXGapProbFcn == emissions_kmer_getGapProb &&
               s->getYGapProbFcn == emissions_signal_strawManGetKmerEventMatchProb &&
               s->getMatchProbFcn == emissions_signal_strawManGetKmerEventMatchProb;
    }
    if (cell == (const void *) sm3hdp_cell) {
        const StateMachine3_HDP *s = (StateMachine3_HDP *) sM;
        return s->getYGapProbFcn == get_nanopore_kmer_density && s->getMatchProbFcn == get_nanopore_kmer_density;
    }
    if (cell == (const void *) vanilla_cell) {
        const StateMachine3Vanilla *s = (StateMachine3Vanilla *) sM;
        return s->getKmerSkipProb == emissions_signal_getBetaOrAlphaSkipProb &&
               s->getScaledMatchProbFcn == emissions_signal_getEventMatchProbWithTwoDists &&
               s->getMatchProbFcn == emissions_signal_getEventMatchProbWithTwoDists;
    }
    return 0;
}

/* ---- constructors taking plug-ins (impl/stateMachine.c:1462-1600) --------------------------------------------- */
StateMachine *stateMachine3_construct(StateMachineType type, int64_t parameterSetSize,
                                      void (*setTransitionsToDefaults)(StateMachine *sM),
                                      void (*setEmissionsDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                      double (*gapXProbFcn)(const double *, void *),
                                      double (*gapYProbFcn)(const double *, void *, void *),
                                      double (*matchProbFcn)(const double *, void *, void *),
                                      void (*cellCalcUpdateExpFcn)(double *, double *, int64_t, int64_t, double, double,
                                                                   void *)) {
    if (type != threeState && type != threeStateAsymmetric)
        die("Tried to create a three state state-machine with the wrong type");
    StateMachine3 *s = calloc(1, sizeof *s);
    s->model.type = type;
    s->model.parameterSetSize = parameterSetSize;
    s->model.stateNumber = 3;
    s->model.matchState = match;
    cpecan_sm3_set_functions(s);
    s->model.cellCalculateUpdateExpectations = cellCalcUpdateExpFcn;
    s->getXGapProbFcn = gapXProbFcn;
    s->getYGapProbFcn = gapYProbFcn;
    s->getMatchProbFcn = matchProbFcn;
    setTransitionsToDefaults((StateMachine *) s);
    setEmissionsDefaults((StateMachine *) s, parameterSetSize);
    for (int64_t i = 0; i < parameterSetSize; i++) s->model.EMISSION_GAP_X_PROBS[i] = -2.3025850929940455;
    return (StateMachine *) s;
}
StateMachine *stateMachine3Hdp_construct(StateMachineType type, int64_t parameterSetSize,
                                         void (*setTransitionsToDefaults)(StateMachine *sM),
                                         void (*setEmissionsDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                         NanoporeHDP *hdpModel, double (*gapXProbFcn)(const double *, void *),
                                         double (*gapYProbFcn)(NanoporeHDP *, void *, void *),
                                         double (*matchProbFcn)(NanoporeHDP *, void *, void *),
                                         void (*cellCalcUpdateExpFcn)(double *, double *, int64_t, int64_t, double,
                                                                      double, void *)) {
    if (type != threeStateHdp) die("Tried to create a three state state-machine with the wrong type");
    StateMachine3_HDP *s = calloc(1, sizeof *s);
    s->model.type = type;
    s->model.parameterSetSize = parameterSetSize;
    s->model.stateNumber = 3;
    s->model.matchState = match;
    cpecan_sm3hdp_set_functions(s);
    s->model.cellCalculateUpdateExpectations = cellCalcUpdateExpFcn;
    s->getXGapProbFcn = gapXProbFcn;
    s->getYGapProbFcn = gapYProbFcn;
    s->getMatchProbFcn = matchProbFcn;
    s->hdpModel = hdpModel;
    setTransitionsToDefaults((StateMachine *) s);
    setEmissionsDefaults((StateMachine *) s, parameterSetSize);
    for (int64_t i = 0; i < parameterSetSize; i++) s->model.EMISSION_GAP_X_PROBS[i] = -2.3025850929940455;
    return (StateMachine *) s;
}
StateMachine *stateMachine3Vanilla_construct(StateMachineType type, int64_t parameterSetSize,
                                             void (*setEmissionsDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                             double (*xSkipProbFcn)(StateMachine *, void *, bool),
                                             double (*scaledMatchProbFcn)(const double *, void *, void *),
                                             double (*matchProbFcn)(const double *, void *, void *),
                                             void (*cellCalcUpdateExpFcn)(double *, double *, int64_t, int64_t, double,
                                                                          double, void *)) {
    if (type != vanilla) die("Tried to create a vanilla state machine with the wrong type?");
    StateMachine3Vanilla *s = calloc(1, sizeof *s);
    s->TRANSITION_M_TO_Y_NOT_X = 0.17;
    s->TRANSITION_E_TO_E = 0.55f;
    s->DEFAULT_END_MATCH_PROB = -0.23552123624314988;
    s->DEFAULT_END_FROM_X_PROB = -1.6269694202638481;
    s->DEFAULT_END_FROM_Y_PROB = -4.3187242127300092;
    s->model.type = type;
    s->model.parameterSetSize = parameterSetSize;
    s->model.stateNumber = 3;
    s->model.matchState = match;
    cpecan_sm3vanilla_set_functions(s);
    s->model.cellCalculateUpdateExpectations = cellCalcUpdateExpFcn;
    s->getKmerSkipProb = xSkipProbFcn;
    s->getScaledMatchProbFcn = scaledMatchProbFcn;
    s->getMatchProbFcn = matchProbFcn;
    setEmissionsDefaults((StateMachine *) s, 60);
    return (StateMachine *) s;
}

/* ------------------------------------------------------------------------------------------------ */
/* cells (impl/pairwiseAligner.c:357-512)                                                           */
/* ------------------------------------------------------------------------------------------------ */
static void forward_arc(double *fromCells, double *toCells, int64_t from, int64_t to, double eP, double tP, void *x) {
    (void) x;
    toCells[to] = logAdd(toCells[to], fromCells[from] + (eP + tP));
}
static void backward_arc(double *fromCells, double *toCells, int64_t from, int64_t to, double eP, double tP, void *x) {
    (void) x;
    fromCells[from] = logAdd(fromCells[from], toCells[to] + (eP + tP));
}
void cell_calculateForward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                           void *cY, void *extraArgs) {
    sM->cellCalculate(sM, current, lower, middle, upper, cX, cY, forward_arc, extraArgs);
}
void cell_calculateBackward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                            void *cY, void *extraArgs) {
    sM->cellCalculate(sM, current, lower, middle, upper, cX, cY, backward_arc, extraArgs);
}
double cell_dotProduct(double *cell1, double *cell2, int64_t stateNumber) {
    double total = cell1[0] + cell2[0];
    for (int64_t i = 1; i < stateNumber; i++) total = logAdd(total, cell1[i] + cell2[i]);
    return total;
}
double cell_dotProduct2(double *cell, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t)) {
    double total = cell[0] + getStateValue(sM, 0);
    for (int64_t i = 1; i < sM->stateNumber; i++) total = logAdd(total, cell[i] + getStateValue(sM, i));
    return total;
}

/* the four update-expectations plug-ins; extraArgs = { &totalProbability, Hmm *, cX, cY } (:500-512) */
static double arc_posterior(double *fromCells, double *toCells, int64_t from, int64_t to, double eP, double tP,
                            void **args) {
    return exp(fromCells[from] + toCells[to] + (eP + tP) - *(double *) args[0]);
}
void cell_updateExpectations(double *fromCells, double *toCells, int64_t from, int64_t to, double eP, double tP,
                             void *extraArgs) {
    void **args = extraArgs;
    Hmm *hmm = args[1];
    const int64_t x = hmm->getElementIndexFcn(args[2]), y = hmm->getElementIndexFcn(args[3]);
    const double p = arc_posterior(fromCells, toCells, from, to, eP, tP, args);
    hmm->addToTransitionExpectationFcn(hmm, from, to, p);
    if (x < hmm->symbolSetSize && y < hmm->symbolSetSize) hmm->addToEmissionExpectationFcn(hmm, to, x, y, p);
}
void cell_signal_updateTransAndKmerSkipExpectations(double *fromCells, double *toCells, int64_t from, int64_t to,
                                                    double eP, double tP, void *extraArgs) {
    void **args = extraArgs;
    Hmm *hmm = args[1];
    const int64_t x = hmm->getElementIndexFcn(args[2]);
    const double p = arc_posterior(fromCells, toCells, from, to, eP, tP, args);
    hmm->addToTransitionExpectationFcn(hmm, from, to, p);
    if (to == shortGapX) hmm->addToEmissionExpectationFcn(hmm, 0, x, 0, p);
}
void cell_signal_updateTransAndKmerSkipExpectations2(double *fromCells, double *toCells, int64_t from, int64_t to,
                                                     double eP, double tP, void *extraArgs) {
    void **args = extraArgs;
    HdpHmm *hmm = args[1];
    const double p = arc_posterior(fromCells, toCells, from, to, eP, tP, args);
    hmm->baseHmm.addToTransitionExpectationFcn((Hmm *) hmm, from, to, p);
    if (to == match && p >= hmm->threshold) hmm->addToAssignments((Hmm *) hmm, args[2], args[3]);
}
void cell_signal_updateBetaAndAlphaProb(double *fromCells, double *toCells, int64_t from, int64_t to, double eP,
                                        double tP, void *extraArgs) {
    void **args = extraArgs;
    VanillaHmm *hmm = args[1];
    Hmm *base = (Hmm *) hmm;
    const int64_t bin = hmm->getKmerSkipBin(hmm->matchModel, args[2]);
    const double p = arc_posterior(fromCells, toCells, from, to, eP, tP, args);
    if (from == match && to == shortGapX) base->addToTransitionExpectationFcn(base, bin, 0, p);
    if (from == shortGapX && to == shortGapX) base->addToTransitionExpectationFcn(base, bin + 30, 0, p);
}
static void cell_update_expectation(StateMachine *sM, double *current, double *lower, double *middle, double *upper,
                                    void *cX, void *cY, void *extraArgs) {
    void *args[4] = { ((void **) extraArgs)[0], ((void **) extraArgs)[1], cX, cY };
    sM->cellCalculate(sM, current, lower, middle, upper, cX, cY, sM->cellCalculateUpdateExpectations, args);
}

/* ------------------------------------------------------------------------------------------------ */
/* DpDiagonal, DpMatrix (impl/pairwiseAligner.c:514-680)                                            */
/* ------------------------------------------------------------------------------------------------ */
struct _dpDiagonal {
    Diagonal diagonal;
    int64_t stateNumber;
    double *cells;
};
static int64_t cell_count(const DpDiagonal *d) { return diagonal_getWidth(d->diagonal) * d->stateNumber; }

DpDiagonal *dpDiagonal_construct(Diagonal diagonal, int64_t stateNumber) {
    DpDiagonal *d = malloc(sizeof *d);
    d->diagonal = diagonal;
    d->stateNumber = stateNumber;
    d->cells = malloc(sizeof(double) * (size_t) (stateNumber * diagonal_getWidth(diagonal)));
    return d;
}
DpDiagonal *dpDiagonal_clone(DpDiagonal *diagonal) {
    DpDiagonal *d = dpDiagonal_construct(diagonal->diagonal, diagonal->stateNumber);
    memcpy(d->cells, diagonal->cells, sizeof(double) * (size_t) cell_count(diagonal));
    return d;
}
bool dpDiagonal_equals(DpDiagonal *a, DpDiagonal *b) {
    if (!diagonal_equals(a->diagonal, b->diagonal) || a->stateNumber != b->stateNumber) return 0;
    for (int64_t i = 0; i < cell_count(a); i++)
        if (a->cells[i] != b->cells[i]) return 0;
    return 1;
}
void dpDiagonal_destruct(DpDiagonal *d) {
    free(d->cells);
    free(d);
}
double *dpDiagonal_getCell(DpDiagonal *d, int64_t xmy) {
    if (xmy < d->diagonal.xmyL || xmy > d->diagonal.xmyR) return NULL;
    assert((d->diagonal.xay + xmy) % 2 == 0);
    return d->cells + ((xmy - d->diagonal.xmyL) / 2) * d->stateNumber;
}
void dpDiagonal_zeroValues(DpDiagonal *d) {
    for (int64_t i = 0; i < cell_count(d); i++) d->cells[i] = LOG_ZERO;
}
void dpDiagonal_initialiseValues(DpDiagonal *d, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t)) {
    for (int64_t xmy = d->diagonal.xmyL; xmy <= d->diagonal.xmyR; xmy += 2) {
        double *cell = dpDiagonal_getCell(d, xmy);
        for (int64_t s = 0; s < d->stateNumber; s++) cell[s] = getStateValue(sM, s);
    }
}
double dpDiagonal_dotProduct(DpDiagonal *a, DpDiagonal *b) {
    double total = LOG_ZERO;
    for (int64_t xmy = a->diagonal.xmyL; xmy <= a->diagonal.xmyR; xmy += 2)
        total = logAdd(total, cell_dotProduct(dpDiagonal_getCell(a, xmy), dpDiagonal_getCell(b, xmy), a->stateNumber));
    return total;
}

struct _dpMatrix {
    DpDiagonal **diagonals;
    int64_t diagonalNumber, activeDiagonals, stateNumber;
};
DpMatrix *dpMatrix_construct(int64_t diagonalNumber, int64_t stateNumber) {
    DpMatrix *m = malloc(sizeof *m);
    m->diagonalNumber = diagonalNumber;
    m->diagonals = calloc((size_t) diagonalNumber + 1, sizeof(DpDiagonal *));
    m->activeDiagonals = 0;
    m->stateNumber = stateNumber;
    return m;
}
void dpMatrix_destruct(DpMatrix *m) {
    assert(m->activeDiagonals == 0);
    free(m->diagonals);
    free(m);
}
DpDiagonal *dpMatrix_getDiagonal(DpMatrix *m, int64_t xay) {
    return xay < 0 || xay > m->diagonalNumber ? NULL : m->diagonals[xay];
}
int64_t dpMatrix_getActiveDiagonalNumber(DpMatrix *m) { return m->activeDiagonals; }
DpDiagonal *dpMatrix_createDiagonal(DpMatrix *m, Diagonal diagonal) {
    if (diagonal.xay < 0 || diagonal.xay > m->diagonalNumber || m->diagonals[diagonal.xay])
        die("cpecan: dpMatrix_createDiagonal: diagonal %lld of a matrix of %lld", (long long) diagonal.xay,
            (long long) m->diagonalNumber);
    m->activeDiagonals++;
    return m->diagonals[diagonal.xay] = dpDiagonal_construct(diagonal, m->stateNumber);
}
void dpMatrix_deleteDiagonal(DpMatrix *m, int64_t xay) {
    if (xay < 0 || xay > m->diagonalNumber) die("cpecan: dpMatrix_deleteDiagonal: diagonal %lld", (long long) xay);
    if (!m->diagonals[xay]) return;
    m->activeDiagonals--;
    dpDiagonal_destruct(m->diagonals[xay]);
    m->diagonals[xay] = NULL;
}

/* ------------------------------------------------------------------------------------------------ */
/* one diagonal at a time (impl/pairwiseAligner.c:682-866)                                          */
/* ------------------------------------------------------------------------------------------------ */
typedef void (*CellFn)(StateMachine *, double *, double *, double *, double *, void *, void *, void *);
static void sweep_diagonal(StateMachine *sM, DpDiagonal *d, DpDiagonal *dM1, DpDiagonal *dM2, Sequence *sX,
                           Sequence *sY, CellFn cellFn, void *extraArgs) {
    const int64_t xay = d->diagonal.xay;
    for (int64_t xmy = d->diagonal.xmyL; xmy <= d->diagonal.xmyR; xmy += 2) {
        void *x = sX->get(sX->elements, diagonal_getXCoordinate(xay, xmy) - 1);
        void *y = sY->get(sY->elements, diagonal_getYCoordinate(xay, xmy) - 1);
        cellFn(sM, dpDiagonal_getCell(d, xmy), dM1 ? dpDiagonal_getCell(dM1, xmy - 1) : NULL,
               dM2 ? dpDiagonal_getCell(dM2, xmy) : NULL, dM1 ? dpDiagonal_getCell(dM1, xmy + 1) : NULL, x, y,
               extraArgs);
    }
}
void diagonalCalculationForward(StateMachine *sM, int64_t xay, DpMatrix *m, Sequence *sX, Sequence *sY) {
    sweep_diagonal(sM, dpMatrix_getDiagonal(m, xay), dpMatrix_getDiagonal(m, xay - 1), dpMatrix_getDiagonal(m, xay - 2),
                   sX, sY, cell_calculateForward, NULL);
}
void diagonalCalculationBackward(StateMachine *sM, int64_t xay, DpMatrix *m, Sequence *sX, Sequence *sY) {
    sweep_diagonal(sM, dpMatrix_getDiagonal(m, xay), dpMatrix_getDiagonal(m, xay - 1), dpMatrix_getDiagonal(m, xay - 2),
                   sX, sY, cell_calculateBackward, NULL);
}
double diagonalCalculationTotalProbability(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                           DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY) {
    double total = dpDiagonal_dotProduct(dpMatrix_getDiagonal(forwardDpMatrix, xay),
                                         dpMatrix_getDiagonal(backwardDpMatrix, xay));
    /* the paths that jump over diagonal xay with a match: forward diagonal xay - 1 pushed one match step on */
    DpDiagonal *f = dpMatrix_getDiagonal(forwardDpMatrix, xay - 1), *b = dpMatrix_getDiagonal(backwardDpMatrix, xay + 1);
    if (f && b) {
        DpDiagonal *stepped = dpDiagonal_clone(b);
        dpDiagonal_zeroValues(stepped);
        sweep_diagonal(sM, stepped, NULL, f, sX, sY, cell_calculateForward, NULL);
        total = logAdd(total, dpDiagonal_dotProduct(stepped, b));
        dpDiagonal_destruct(stepped);
    }
    return total;
}
void diagonalCalculationPosteriorMatchProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                            DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY,
                                            double totalProbability, PairwiseAlignmentParameters *p, void *extraArgs) {
    (void) sX; (void) sY;
    stList *alignedPairs = ((void **) extraArgs)[0];
    DpDiagonal *f = dpMatrix_getDiagonal(forwardDpMatrix, xay), *b = dpMatrix_getDiagonal(backwardDpMatrix, xay);
    for (int64_t xmy = f->diagonal.xmyL; xmy <= f->diagonal.xmyR; xmy += 2) {
        const int64_t x = diagonal_getXCoordinate(xay, xmy), y = diagonal_getYCoordinate(xay, xmy);
        if (x <= 0 || y <= 0) continue;
        double posterior = exp(dpDiagonal_getCell(f, xmy)[sM->matchState] + dpDiagonal_getCell(b, xmy)[sM->matchState] -
                               totalProbability);
        if (posterior < p->threshold) continue;
        if (posterior > 1.0) posterior = 1.0;
        stList_append(alignedPairs, stIntTuple_construct3((int64_t) floor(posterior * PAIR_ALIGNMENT_PROB_1), x - 1, y - 1));
    }
}
/* the echelon machine's decode (:797-839): states matchState .. 5 of a cell, state s standing for s k-mers.  Host
 * only, like every function of this block; the echelon machine itself is not built (the reference marks it broken). */
void diagonalCalculationMultiPosteriorMatchProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                                 DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY,
                                                 double totalProbability, PairwiseAlignmentParameters *p,
                                                 void *extraArgs) {
    (void) sX; (void) sY;
    stList *alignedPairs = ((void **) extraArgs)[0];
    DpDiagonal *f = dpMatrix_getDiagonal(forwardDpMatrix, xay), *b = dpMatrix_getDiagonal(backwardDpMatrix, xay);
    for (int64_t xmy = f->diagonal.xmyL; xmy <= f->diagonal.xmyR; xmy += 2) {
        const int64_t x = diagonal_getXCoordinate(xay, xmy), y = diagonal_getYCoordinate(xay, xmy);
        if (x <= 0 || y <= 0) continue;
        for (int64_t s = sM->matchState; s < 6 && s < sM->stateNumber; s++) {
            double posterior = exp(dpDiagonal_getCell(f, xmy)[s] + dpDiagonal_getCell(b, xmy)[s] - totalProbability);
            if (posterior < p->threshold) continue;
            if (posterior > 1.0) posterior = 1.0;
            for (int64_t n = 0; n < s; n++)
                stList_append(alignedPairs,
                              stIntTuple_construct3((int64_t) floor(posterior * PAIR_ALIGNMENT_PROB_1), x + n - 1, y - 1));
        }
    }
}
void diagonalCalculation_Expectations(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                      DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY, double totalProbability,
                                      PairwiseAlignmentParameters *p, void *extraArgs) {
    (void) p;
    Hmm *hmm = extraArgs;
    void *args[2] = { &totalProbability, hmm };
    hmm->likelihood += totalProbability;
    sweep_diagonal(sM, dpMatrix_getDiagonal(backwardDpMatrix, xay), dpMatrix_getDiagonal(forwardDpMatrix, xay - 1),
                   dpMatrix_getDiagonal(forwardDpMatrix, xay - 2), sX, sY, cell_update_expectation, args);
}

/* ------------------------------------------------------------------------------------------------ */
/* Hmm subclasses of the signal machines (impl/continuousHmm.c)                                     */
/* ------------------------------------------------------------------------------------------------ */
static void hmm_base_init(Hmm *h, StateMachineType type, int64_t stateNumber, int64_t symbolSetSize) {
    memset(h, 0, sizeof *h);
    h->type = type;
    h->stateNumber = stateNumber;
    h->symbolSetSize = symbolSetSize;
    h->matrixSize = MODEL_PARAMS;
    h->likelihood = 0.0;
}
static double uniform01(void) { return rand() / ((double) RAND_MAX + 1.0); } /* st_random() */

/* ---- ContinuousPairHmm :86-370 ---- */
Hmm *continuousPairHmm_constructEmpty(
    double pseudocount, int64_t stateNumber, int64_t symbolSetSize, StateMachineType type,
    void (*addToTransitionExpFcn)(Hmm *, int64_t, int64_t, double), void (*setTransitionFcn)(Hmm *, int64_t, int64_t, double),
    double (*getTransitionsExpFcn)(Hmm *, int64_t, int64_t),
    void (*addToKmerGapExpFcn)(Hmm *, int64_t, int64_t, int64_t, double),
    void (*setKmerGapExpFcn)(Hmm *, int64_t, int64_t, int64_t, double),
    double (*getKmerGapExpFcn)(Hmm *, int64_t, int64_t, int64_t), int64_t (*getElementIndexFcn)(void *)) {
    if (type != threeState && type != threeStateHdp)
        die("ContinuousPair HMM construct: Wrong HMM type for this function got: %i", (int) type);
    ContinuousPairHmm *c = malloc(sizeof *c);
    Hmm *h = (Hmm *) c;
    hmm_base_init(h, type, stateNumber, symbolSetSize);
    h->addToTransitionExpectationFcn = addToTransitionExpFcn;
    h->setTransitionFcn = setTransitionFcn;
    h->getTransitionsExpFcn = getTransitionsExpFcn;
    h->addToEmissionExpectationFcn = addToKmerGapExpFcn;
    h->setEmissionExpectationFcn = setKmerGapExpFcn;
    h->getEmissionExpFcn = getKmerGapExpFcn;
    h->getElementIndexFcn = getElementIndexFcn;
    c->transitions = malloc(sizeof(double) * (size_t) (stateNumber * stateNumber));
    for (int64_t i = 0; i < stateNumber * stateNumber; i++) c->transitions[i] = pseudocount;
    c->individualKmerGapProbs = malloc(sizeof(double) * (size_t) symbolSetSize);
    for (int64_t i = 0; i < symbolSetSize; i++) c->individualKmerGapProbs[i] = pseudocount;
    return h;
}
void continuousPairHmm_addToTransitionsExpectation(Hmm *hmm, int64_t from, int64_t to, double p) {
    ((ContinuousPairHmm *) hmm)->transitions[from * hmm->stateNumber + to] += p;
}
void continuousPairHmm_setTransitionExpectation(Hmm *hmm, int64_t from, int64_t to, double p) {
    ((ContinuousPairHmm *) hmm)->transitions[from * hmm->stateNumber + to] = p;
}
double continuousPairHmm_getTransitionExpectation(Hmm *hmm, int64_t from, int64_t to) {
    return ((ContinuousPairHmm *) hmm)->transitions[from * hmm->stateNumber + to];
}
void continuousPairHmm_addToKmerGapExpectation(Hmm *hmm, int64_t state, int64_t kmerIndex, int64_t ignore, double p) {
    (void) state; (void) ignore;
    ((ContinuousPairHmm *) hmm)->individualKmerGapProbs[kmerIndex] += p;
}
void continuousPairHmm_setKmerGapExpectation(Hmm *hmm, int64_t state, int64_t kmerIndex, int64_t ignore, double p) {
    (void) state; (void) ignore;
    ((ContinuousPairHmm *) hmm)->individualKmerGapProbs[kmerIndex] = p;
}
double continuousPairHmm_getKmerGapExpectation(Hmm *hmm, int64_t state, int64_t kmerIndex, int64_t ignore) {
    (void) state; (void) ignore;
    return ((ContinuousPairHmm *) hmm)->individualKmerGapProbs[kmerIndex];
}
void continuousPairHmm_destruct(Hmm *hmm) {
    ContinuousPairHmm *c = (ContinuousPairHmm *) hmm;
    free(c->transitions);
    free(c->individualKmerGapProbs);
    free(c);
}
void continuousPairHmm_normalize(Hmm *hmm) {
    if (hmm->type != threeState) die("continuousPairHmm_normalize: got invalid HMM type: %lld", (long long) hmm->type);
    hmmDiscrete_normalize2(hmm, 0);
    double total = 0.0;
    for (int64_t i = 0; i < hmm->symbolSetSize; i++) total += hmm->getEmissionExpFcn(hmm, 0, i, 0);
    for (int64_t i = 0; i < hmm->symbolSetSize; i++)
        hmm->setEmissionExpectationFcn(hmm, 0, i, 0, hmm->getEmissionExpFcn(hmm, 0, i, 0) / total);
}
void continuousPairHmm_randomize(Hmm *hmm) {
    for (int64_t from = 0; from < hmm->stateNumber; from++)
        for (int64_t to = 0; to < hmm->stateNumber; to++) hmm->setTransitionFcn(hmm, from, to, uniform01());
    for (int64_t i = 0; i < hmm->symbolSetSize; i++) hmm->setEmissionExpectationFcn(hmm, 0, i, 0, uniform01());
    continuousPairHmm_normalize(hmm);
}
/* the three-state M-step shared by the strawMan and HDP machines (their nine transitions sit at the same offsets) */
static void load_three_state_transitions(StateMachine3 *s, Hmm *hmm) {
    s->TRANSITION_MATCH_CONTINUE = log(hmm->getTransitionsExpFcn(hmm, match, match));
    s->TRANSITION_GAP_OPEN_X = log(hmm->getTransitionsExpFcn(hmm, match, shortGapX));
    s->TRANSITION_GAP_OPEN_Y = log(hmm->getTransitionsExpFcn(hmm, match, shortGapY));
    s->TRANSITION_MATCH_FROM_GAP_X = log(hmm->getTransitionsExpFcn(hmm, shortGapX, match));
    s->TRANSITION_GAP_EXTEND_X = log(1 - hmm->getTransitionsExpFcn(hmm, shortGapX, match)); /* tied */
    s->TRANSITION_GAP_SWITCH_TO_Y = LOG_ZERO;
    s->TRANSITION_MATCH_FROM_GAP_Y = log(hmm->getTransitionsExpFcn(hmm, shortGapY, match));
    s->TRANSITION_GAP_EXTEND_Y = log(hmm->getTransitionsExpFcn(hmm, shortGapY, shortGapY));
    s->TRANSITION_GAP_SWITCH_TO_X = log(hmm->getTransitionsExpFcn(hmm, shortGapY, shortGapX));
}
void continuousPairHmm_loadTransitionsAndKmerGapProbs(StateMachine *sM, Hmm *hmm) {
    load_three_state_transitions((StateMachine3 *) sM, hmm);
    for (int64_t i = 0; i < hmm->symbolSetSize; i++) sM->EMISSION_GAP_X_PROBS[i] = log(hmm->getEmissionExpFcn(hmm, 0, i, 0));
}
static void pair_hmm_to_plain(Hmm *hmm, ContinuousPairHmmExpectations *e) {
    if (hmm->stateNumber != 3 || hmm->symbolSetSize != NUM_OF_KMERS)
        die("cpecan: the .hmm file of the strawMan machine holds 3 states and %d k-mers", NUM_OF_KMERS);
    e->likelihood = hmm->likelihood;
    for (int64_t i = 0; i < 9; i++) e->transitions[i] = hmm->getTransitionsExpFcn(hmm, i / 3, i % 3);
    for (int64_t i = 0; i < NUM_OF_KMERS; i++) e->individualKmerGapProbs[i] = hmm->getEmissionExpFcn(hmm, 0, i, 0);
}
void continuousPairHmm_writeToFile(Hmm *hmm, FILE *fileHandle) {
    ContinuousPairHmmExpectations *e = calloc(1, sizeof *e);
    pair_hmm_to_plain(hmm, e);
    cpecan_pairHmmExpectations_write(e, fileHandle);
    free(e);
}
Hmm *continuousPairHmm_loadFromFile(const char *fileName) {
    ContinuousPairHmmExpectations *e = cpecan_pairHmmExpectations_read(fileName);
    Hmm *hmm = hmmContinuous_getEmptyHmm(threeState, 0.0, 0.0);
    hmm->likelihood = e->likelihood;
    for (int64_t i = 0; i < 9; i++) hmm->setTransitionFcn(hmm, i / 3, i % 3, e->transitions[i]);
    for (int64_t i = 0; i < NUM_OF_KMERS; i++) hmm->setEmissionExpectationFcn(hmm, 0, i, 0, e->individualKmerGapProbs[i]);
    free(e);
    return hmm;
}

/* ---- VanillaHmm :373-626 ---- */
#define VANILLA_TABLE (1 + (size_t) NUM_OF_KMERS * MODEL_PARAMS)
Hmm *vanillaHmm_constructEmpty(double pseudocount, int64_t stateNumber, int64_t symbolSetSize, StateMachineType type,
                               void (*addToKmerBinExpFcn)(Hmm *, int64_t, int64_t, double),
                               void (*setKmerBinFcn)(Hmm *, int64_t, int64_t, double),
                               double (*getKmerBinExpFcn)(Hmm *, int64_t, int64_t)) {
    if (type != vanilla) die("Vanilla HMM construct: Wrong HMM type for this function got: %i", (int) type);
    VanillaHmm *v = malloc(sizeof *v);
    Hmm *h = (Hmm *) v;
    hmm_base_init(h, type, stateNumber, symbolSetSize);
    h->addToTransitionExpectationFcn = addToKmerBinExpFcn;
    h->setTransitionFcn = setKmerBinFcn;
    h->getTransitionsExpFcn = getKmerBinExpFcn;
    v->kmerSkipBins = malloc(sizeof(double) * 60);
    for (int i = 0; i < 60; i++) v->kmerSkipBins[i] = pseudocount;
    const size_t table = 1 + (size_t) symbolSetSize * MODEL_PARAMS;
    v->matchModel = calloc(table, sizeof(double));
    v->scaledMatchModel = calloc(table, sizeof(double));
    v->getKmerSkipBin = emissions_signal_getKmerSkipBin;
    return h;
}
void vanillaHmm_addToKmerSkipBinExpectation(Hmm *hmm, int64_t bin, int64_t ignore, double p) {
    (void) ignore;
    ((VanillaHmm *) hmm)->kmerSkipBins[bin] += p;
}
void vanillaHmm_setKmerSkipBinExpectation(Hmm *hmm, int64_t bin, int64_t ignore, double p) {
    (void) ignore;
    ((VanillaHmm *) hmm)->kmerSkipBins[bin] = p;
}
double vanillaHmm_getKmerSkipBinExpectation(Hmm *hmm, int64_t bin, int64_t ignore) {
    (void) ignore;
    return ((VanillaHmm *) hmm)->kmerSkipBins[bin];
}
void vanillaHmm_normalizeKmerSkipBins(Hmm *hmm) { /* alpha and beta bins together, as the reference does */
    double total = 0.0;
    for (int64_t i = 0; i < 60; i++) total += hmm->getTransitionsExpFcn(hmm, i, 0);
    for (int64_t i = 0; i < 60; i++) hmm->setTransitionFcn(hmm, i, 0, hmm->getTransitionsExpFcn(hmm, i, 0) / total);
}
void vanillaHmm_randomizeKmerSkipBins(Hmm *hmm) {
    for (int64_t i = 0; i < 60; i++) hmm->setTransitionFcn(hmm, i, 0, uniform01());
    vanillaHmm_normalizeKmerSkipBins(hmm);
}
void vanillaHmm_implantMatchModelsintoHmm(StateMachine *sM, Hmm *hmm) {
    VanillaHmm *v = (VanillaHmm *) hmm;
    const size_t table = 1 + (size_t) sM->parameterSetSize * MODEL_PARAMS;
    memcpy(v->matchModel, sM->EMISSION_MATCH_PROBS, sizeof(double) * table);
    memcpy(v->scaledMatchModel, sM->EMISSION_GAP_Y_PROBS, sizeof(double) * table);
}
void vanillaHmm_loadKmerSkipBinExpectations(StateMachine *sM, Hmm *hmm) {
    if (hmm->type != vanilla) die("you gave me the wrong type of HMM");
    for (int64_t i = 0; i < 60; i++) sM->EMISSION_GAP_X_PROBS[i] = hmm->getTransitionsExpFcn(hmm, i, 0);
}
void vanillaHmm_destruct(Hmm *hmm) {
    VanillaHmm *v = (VanillaHmm *) hmm;
    free(v->matchModel);
    free(v->scaledMatchModel);
    free(v->kmerSkipBins);
    free(v);
}
/* the tables travel through a StateMachine-shaped view of the Hmm's two tables */
static StateMachine table_view(VanillaHmm *v) {
    StateMachine view;
    memset(&view, 0, sizeof view);
    view.type = vanilla;
    view.parameterSetSize = NUM_OF_KMERS;
    view.EMISSION_MATCH_PROBS = v->matchModel;
    view.EMISSION_GAP_Y_PROBS = v->scaledMatchModel;
    return view;
}
void vanillaHmm_writeToFile(Hmm *hmm, FILE *fileHandle) {
    VanillaHmm *v = (VanillaHmm *) hmm;
    if (hmm->symbolSetSize != NUM_OF_KMERS) die("cpecan: the vanilla .hmm file holds %d k-mers", NUM_OF_KMERS);
    VanillaHmmExpectations e;
    e.likelihood = hmm->likelihood;
    for (int64_t i = 0; i < 60; i++) e.kmerSkipBins[i] = hmm->getTransitionsExpFcn(hmm, i, 0);
    StateMachine view = table_view(v);
    cpecan_vanillaExpectations_write(&e, &view, fileHandle);
}
Hmm *vanillaHmm_loadFromFile(const char *fileName) {
    Hmm *hmm = hmmContinuous_getEmptyHmm(vanilla, 0.0, 0.0);
    StateMachine view = table_view((VanillaHmm *) hmm);
    VanillaHmmExpectations *e = cpecan_vanillaExpectations_read(fileName, &view);
    hmm->likelihood = e->likelihood;
    for (int64_t i = 0; i < 60; i++) hmm->setTransitionFcn(hmm, i, 0, e->kmerSkipBins[i]);
    free(e);
    return hmm;
}

/* ---- HdpHmm :631-900 ---- */
/* The assignment lists hold the pointers addToAssignments was given -- into the caller's sequences on the E-step, as
 * in the reference (impl/pairwiseAligner.c:445-476) -- and never free them; what hdpHmm_loadFromFile reads from a file
 * is owned by the box around the HdpHmm. */
typedef struct {
    HdpHmm hmm;
    double *ownedEvents;
    char *ownedKmers;
} HdpHmmBox;
static void hdp_add_assignment(Hmm *self, void *kmerPtr, void *eventPtr) {
    HdpHmm *h = (HdpHmm *) self;
    stList_append(h->kmerAssignments, kmerPtr);
    stList_append(h->eventAssignments, eventPtr);
    h->numberOfAssignments += 1;
}
Hmm *hdpHmm_constructEmpty(double pseudocount, int64_t stateNumber, StateMachineType type, double threshold,
                           void (*addToTransitionExpFcn)(Hmm *, int64_t, int64_t, double),
                           void (*setTransitionFcn)(Hmm *, int64_t, int64_t, double),
                           double (*getTransitionsExpFcn)(Hmm *, int64_t, int64_t)) {
    HdpHmmBox *box = calloc(1, sizeof *box);
    HdpHmm *h = &box->hmm;
    hmm_base_init(&h->baseHmm, type, stateNumber, 0);
    h->baseHmm.addToTransitionExpectationFcn = addToTransitionExpFcn;
    h->baseHmm.setTransitionFcn = setTransitionFcn;
    h->baseHmm.getTransitionsExpFcn = getTransitionsExpFcn;
    h->transitions = malloc(sizeof(double) * (size_t) (stateNumber * stateNumber));
    for (int64_t i = 0; i < stateNumber * stateNumber; i++) h->transitions[i] = pseudocount;
    h->threshold = threshold;
    h->addToAssignments = hdp_add_assignment;
    h->kmerAssignments = stList_construct();
    h->eventAssignments = stList_construct();
    h->numberOfAssignments = 0;
    h->nhdp = NULL;
    return (Hmm *) h;
}
void hdpHmm_loadTransitions(StateMachine *sM, Hmm *hmm) { load_three_state_transitions((StateMachine3 *) sM, hmm); }
void hdpHmm_writeToFile(Hmm *hmm, FILE *fileHandle) {
    HdpHmm *h = (HdpHmm *) hmm;
    if (stList_length(h->kmerAssignments) != stList_length(h->eventAssignments)) return; /* hdpHmm_checkAssignments */
    HdpHmmExpectations *e = cpecan_hdpExpectations_construct(0.0, h->threshold);
    e->likelihood = hmm->likelihood;
    for (int64_t i = 0; i < 9; i++) e->transitions[i] = hmm->getTransitionsExpFcn(hmm, i / 3, i % 3);
    const int64_t n = h->numberOfAssignments;
    e->capacity = n > 0 ? n : 1;
    e->eventAssignments = malloc(sizeof(double) * (size_t) e->capacity);
    e->kmerAssignments = calloc((size_t) e->capacity, KMER_LENGTH + 1);
    for (int64_t i = 0; i < n; i++) {
        e->eventAssignments[i] = *(double *) stList_get(h->eventAssignments, i);
        memcpy(e->kmerAssignments + i * (KMER_LENGTH + 1), stList_get(h->kmerAssignments, i), KMER_LENGTH);
    }
    e->numberOfAssignments = n;
    cpecan_hdpExpectations_write(e, fileHandle);
    cpecan_hdpExpectations_destruct(e);
}
Hmm *hdpHmm_loadFromFile2(const char *fileName, NanoporeHDP *nHdp) { return hdpHmm_loadFromFile(fileName, nHdp); }
Hmm *hdpHmm_loadFromFile(const char *fileName, NanoporeHDP *nHdp) {
    HdpHmmExpectations *e = cpecan_hdpExpectations_read(fileName);
    Hmm *hmm = hmmContinuous_getEmptyHmm(threeStateHdp, 0.0, e->threshold);
    HdpHmmBox *box = (HdpHmmBox *) hmm;
    hmm->likelihood = e->likelihood;
    for (int64_t i = 0; i < 9; i++) hmm->setTransitionFcn(hmm, i / 3, i % 3, e->transitions[i]);
    /* the reference hands the assignments to the Gibbs sampler of nHdp here (:860-893); resampling an HDP is not on
     * this path, so they are kept on the Hmm for the caller instead */
    box->ownedEvents = e->eventAssignments;
    box->ownedKmers = e->kmerAssignments;
    for (int64_t i = 0; i < e->numberOfAssignments; i++)
        box->hmm.addToAssignments(hmm, box->ownedKmers + i * (KMER_LENGTH + 1), box->ownedEvents + i);
    box->hmm.nhdp = nHdp;
    free(e);
    return hmm;
}
void hdpHmm_destruct(Hmm *hmm) {
    HdpHmmBox *box = (HdpHmmBox *) hmm;
    stList_destruct(box->hmm.kmerAssignments);
    stList_destruct(box->hmm.eventAssignments);
    free(box->hmm.transitions);
    free(box->ownedEvents);
    free(box->ownedKmers);
    free(box);
}

/* ---- hmmContinuous_* :903-987 ---- */
static void check_signal_type(const char *who, StateMachineType type) {
    if (type != threeStateHdp && type != threeState && type != vanilla)
        die("%s - ERROR: got unsupported HMM type %i", who, (int) type);
}
Hmm *hmmContinuous_getEmptyHmm(StateMachineType type, double pseudocount, double threshold) {
    check_signal_type("hmmContinuous_getEmptyHmm", type);
    if (type == vanilla)
        return vanillaHmm_constructEmpty(pseudocount, 3, NUM_OF_KMERS, vanilla, vanillaHmm_addToKmerSkipBinExpectation,
                                         vanillaHmm_setKmerSkipBinExpectation, vanillaHmm_getKmerSkipBinExpectation);
    if (type == threeState)
        return continuousPairHmm_constructEmpty(
            pseudocount, 3, NUM_OF_KMERS, threeState, continuousPairHmm_addToTransitionsExpectation,
            continuousPairHmm_setTransitionExpectation, continuousPairHmm_getTransitionExpectation,
            continuousPairHmm_addToKmerGapExpectation, continuousPairHmm_setKmerGapExpectation,
            continuousPairHmm_getKmerGapExpectation, emissions_discrete_getKmerIndex);
    /* the HdpHmm keeps its transitions where the ContinuousPairHmm does (first member after the base), which is
     * what lets the reference reuse these three accessors */
    return hdpHmm_constructEmpty(pseudocount, 3, threeStateHdp, threshold, continuousPairHmm_addToTransitionsExpectation,
                                 continuousPairHmm_setTransitionExpectation, continuousPairHmm_getTransitionExpectation);
}
void hmmContinuous_loadSignalHmm(const char *hmmFile, StateMachine *sM, StateMachineType type) {
    check_signal_type("hmmContinuous_loadSignalHmm", type);
    if (type == vanilla) {
        Hmm *hmm = vanillaHmm_loadFromFile(hmmFile);
        vanillaHmm_loadKmerSkipBinExpectations(sM, hmm);
        vanillaHmm_destruct(hmm);
    } else if (type == threeState) {
        Hmm *hmm = continuousPairHmm_loadFromFile(hmmFile);
        continuousPairHmm_loadTransitionsAndKmerGapProbs(sM, hmm);
        continuousPairHmm_destruct(hmm);
    } else {
        Hmm *hmm = hdpHmm_loadFromFile(hmmFile, NULL);
        hdpHmm_loadTransitions(sM, hmm);
        hdpHmm_destruct(hmm);
    }
}
void hmmContinuous_normalize(Hmm *hmm, StateMachineType type) {
    if (type == vanilla) vanillaHmm_normalizeKmerSkipBins(hmm);
    else if (type == threeState) continuousPairHmm_normalize(hmm);
    else die("hmmContinuous_normalize - ERROR: got unsupported HMM type %i", (int) type);
}
void hmmContinuous_writeToFile(const char *outFile, Hmm *hmm, StateMachineType type) {
    check_signal_type("hmmContinuous_writeToFile", type);
    FILE *fH = fopen(outFile, "w");
    if (!fH) die("cpecan: cannot write %s", outFile);
    if (type == vanilla) vanillaHmm_writeToFile(hmm, fH);
    else if (type == threeState) continuousPairHmm_writeToFile(hmm, fH);
    else hdpHmm_writeToFile(hmm, fH);
    fclose(fH);
}
void hmmContinuous_destruct(Hmm *hmm, StateMachineType type) {
    if (type == vanilla) vanillaHmm_destruct(hmm);
    else if (type == threeState) continuousPairHmm_destruct(hmm);
    else if (type == threeStateHdp) hdpHmm_destruct(hmm);
}
int64_t hmmContinuous_howManyAssignments(Hmm *hmm) {
    if (hmm->type != threeStateHdp)
        die("hmmContinuous: this type of Hmm doesn't have assignments got type: %lld", (long long) hmm->type);
    return ((HdpHmm *) hmm)->numberOfAssignments;
}
