/*
 * cpecan_api.c -- host C code behind include/cpecan_api.h: the reference-shaped entry points of the
 * banded pair-HMM posterior path, implemented on top of the C-ABI of cpecan_hip.h.
 *
 * What runs here is control plane only: argument checking, the reference's split of an alignment at
 * large anchor-free gaps (getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps,
 * impl/pairwiseAligner.c:1356-1422), packing of reads into one GPU batch, and the reference's
 * result-list conventions (coordinate shift + tail-first append, :1342-1354,:1447-1454).  No DP
 * cell is computed on the host; a missing GPU or a C-ABI error aborts with a message, which is the
 * reference's own error convention (st_errAbort).
 */
#include "cpecan_host_private.h"

#include "cpecan_hip.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

void cpecan_die(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    /* st_errAbort's convention: print and leave with status 1.  Other threads of the caller may be inside the GPU
     * runtime at this moment (the reference's callers align two strands at once), so the process leaves without
     * running exit handlers under them */
    fflush(NULL);
    _exit(1);
}
#define die cpecan_die
#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != CPECAN_OK) die("cpecan: %s failed (%d): %s", #call, rc_, cpecan_hip_last_error()); \
    } while (0)

/* ------------------------------------------------------------------------------------------------ */
/* minimal stList / stIntTuple                                                                      */
/* ------------------------------------------------------------------------------------------------ */
#ifndef CPECAN_WITH_SONLIB
struct _stList {
    void **items;
    int64_t n, cap;
    void (*destruct)(void *);
};
struct _stIntTuple {
    int64_t n;
    int64_t v[4];
};
stList *stList_construct3(int64_t size, void (*destructElement)(void *)) {
    stList *l = calloc(1, sizeof(stList));
    l->cap = size > 8 ? size : 8;
    l->items = calloc((size_t) l->cap, sizeof(void *));
    l->n = size;
    l->destruct = destructElement;
    return l;
}
stList *stList_construct(void) { return stList_construct3(0, NULL); }
void stList_destruct(stList *l) {
    if (!l) return;
    if (l->destruct)
        for (int64_t i = 0; i < l->n; i++)
            if (l->items[i]) l->destruct(l->items[i]);
    free(l->items);
    free(l);
}
int64_t stList_length(stList *l) { return l ? l->n : 0; }
void *stList_get(stList *l, int64_t i) { return l->items[i]; }
void stList_append(stList *l, void *item) {
    if (l->n == l->cap) {
        l->cap *= 2;
        l->items = realloc(l->items, (size_t) l->cap * sizeof(void *));
    }
    l->items[l->n++] = item;
}
stIntTuple *stIntTuple_construct2(int64_t a, int64_t b) {
    stIntTuple *t = malloc(sizeof(stIntTuple));
    t->n = 2; t->v[0] = a; t->v[1] = b;
    return t;
}
stIntTuple *stIntTuple_construct3(int64_t a, int64_t b, int64_t c) {
    stIntTuple *t = malloc(sizeof(stIntTuple));
    t->n = 3; t->v[0] = a; t->v[1] = b; t->v[2] = c;
    return t;
}
static stIntTuple *tuple4(int64_t a, int64_t b, int64_t c, int64_t d) {
    stIntTuple *t = malloc(sizeof(stIntTuple));
    t->n = 4; t->v[0] = a; t->v[1] = b; t->v[2] = c; t->v[3] = d;
    return t;
}
/* stList_sort hands the comparison the elements themselves (sonLib's convention), qsort pointers to them */
static __thread int (*sort_cmp)(const void *, const void *);
static int sort_adapter(const void *a, const void *b) { return sort_cmp(*(void *const *) a, *(void *const *) b); }
void stList_sort(stList *l, int (*cmpFn)(const void *a, const void *b)) {
    sort_cmp = cmpFn;
    qsort(l->items, (size_t) l->n, sizeof(void *), sort_adapter);
}
int stIntTuple_cmpFn(const void *a, const void *b) { /* element by element, then by length */
    const stIntTuple *x = a, *y = b;
    for (int64_t i = 0; i < x->n && i < y->n; i++)
        if (x->v[i] != y->v[i]) return x->v[i] < y->v[i] ? -1 : 1;
    return x->n == y->n ? 0 : x->n < y->n ? -1 : 1;
}
int64_t stIntTuple_get(stIntTuple *t, int64_t i) { return t->v[i]; }
int64_t stIntTuple_length(stIntTuple *t) { return t->n; }
void stIntTuple_destruct(stIntTuple *t) { free(t); }
#endif

/* ------------------------------------------------------------------------------------------------ */
/* Sequence, parameters                                                                             */
/* ------------------------------------------------------------------------------------------------ */
static double NULLEVENT_[] = { -INFINITY, 0 };

Sequence *sequence_construct(int64_t length, void *elements, void *(*getFcn)(void *, int64_t)) {
    return sequence_construct2(length, elements, getFcn, NULL);
}
Sequence *sequence_construct2(int64_t length, void *elements, void *(*getFcn)(void *, int64_t),
                              Sequence *(*sliceFcn)(Sequence *, int64_t, int64_t)) {
    Sequence *s = malloc(sizeof(Sequence));
    s->length = length; s->elements = elements; s->get = getFcn; s->sliceFcn = sliceFcn;
    return s;
}
Sequence *sequence_sliceNucleotideSequence2(Sequence *in, int64_t start, int64_t sliceLength) {
    return sequence_construct2(sliceLength, (char *) in->elements + start, in->get, in->sliceFcn);
}
Sequence *sequence_sliceEventSequence2(Sequence *in, int64_t start, int64_t sliceLength) {
    return sequence_construct2(sliceLength, (double *) in->elements + start * NB_EVENT_PARAMS, in->get,
                               in->sliceFcn);
}
void sequence_sequenceDestroy(Sequence *seq) { free(seq); }
void *sequence_getKmer(void *elements, int64_t index) {
    static char n[KMER_LENGTH + 1] = "nnnnnn";
    return index >= 0 ? (void *) &((char *) elements)[index] : (void *) n;
}
void *sequence_getKmer2(void *elements, int64_t index) {
    if (index < 0) return (char *) elements;
    return index > 0 ? (char *) elements + index - 1 : (char *) elements + index;
}
void *sequence_getKmer3(void *elements, int64_t index) {
    return index >= 0 ? (char *) elements + index : (char *) elements;
}
void *sequence_getBase(void *elements, int64_t index) {
    return index >= 0 ? (void *) ((char *) elements + index) : (void *) "n";
}
Sequence *sequence_sliceNucleotideSequence(Sequence *in, int64_t start, int64_t sliceLength) {
    return sequence_construct2(sliceLength, (char *) in->elements + start, in->get, in->sliceFcn);
}
void *sequence_getEvent(void *elements, int64_t index) {
    return index >= 0 ? (void *) &((double *) elements)[index * NB_EVENT_PARAMS] : (void *) NULLEVENT_;
}
int64_t sequence_correctSeqLength(int64_t length, SequenceType type) {
    if (length <= 0) return 0;
    return type == nucleotide ? length : length - (KMER_LENGTH - 1);
}

PairwiseAlignmentParameters *pairwiseAlignmentBandingParameters_construct(void) {
    PairwiseAlignmentParameters *p = malloc(sizeof(PairwiseAlignmentParameters));
    p->threshold = 0.01;
    p->minDiagsBetweenTraceBack = 1000;
    p->traceBackDiagonals = 40;
    p->diagonalExpansion = 20;
    p->constraintDiagonalTrim = 14;
    p->anchorMatrixBiggerThanThis = 500 * 500;
    p->repeatMaskMatrixBiggerThanThis = 500 * 500;
    p->splitMatrixBiggerThanThis = (int64_t) 3000 * 3000;
    p->alignAmbiguityCharacters = 0;
    p->gapGamma = 0.5;
    return p;
}
void pairwiseAlignmentBandingParameters_destruct(PairwiseAlignmentParameters *p) { free(p); }

/* ------------------------------------------------------------------------------------------------ */
/* StateMachine3 (strawMan)                                                                         */
/* ------------------------------------------------------------------------------------------------ */
void stateMachine3_setTransitionsToNanoporeDefaults(StateMachine *sM) {
    StateMachine3 *s = (StateMachine3 *) sM;
    s->TRANSITION_MATCH_CONTINUE = -0.23552123624314988;
    s->TRANSITION_MATCH_FROM_GAP_X = -0.21880828092192281;
    s->TRANSITION_MATCH_FROM_GAP_Y = -0.013406326748077823;
    s->TRANSITION_GAP_OPEN_X = -1.6269694202638481;
    s->TRANSITION_GAP_OPEN_Y = -4.3187242127300092;
    s->TRANSITION_GAP_EXTEND_X = -1.6269694202638481;
    s->TRANSITION_GAP_EXTEND_Y = -4.3187242127239411;
    s->TRANSITION_GAP_SWITCH_TO_X = -INFINITY;
    s->TRANSITION_GAP_SWITCH_TO_Y = -INFINITY;
}

/* ---- 5-state symbol machine (impl/stateMachine.c:896-965, :60-82, :155-173) ------------------- */
static int base_index(void *base) { /* emissions_discrete_getBaseIndex :104-118 */
    switch (*(char *) base) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}
double emissions_symbol_getGapProb(const double *emissionGapProbs, void *base) {
    const int i = base_index(base);
    return i < 4 ? emissionGapProbs[i] : -INFINITY;
}
double emissions_symbol_getMatchProb(const double *emissionMatchProbs, void *x, void *y) {
    const int iX = base_index(x), iY = base_index(y);
    return iX < 4 && iY < 4 ? emissionMatchProbs[iX * 4 + iY] : -INFINITY;
}
void emissions_symbol_setEmissionsToDefaults(StateMachine *sM) {
    const double EMISSION_MATCH = -2.1149196655034745, EMISSION_TRANSVERSION = -4.5691014376830479,
                 EMISSION_TRANSITION = -3.9833860032220842;
    const double M[16] = { EMISSION_MATCH, EMISSION_TRANSVERSION, EMISSION_TRANSITION, EMISSION_TRANSVERSION,
                           EMISSION_TRANSVERSION, EMISSION_MATCH, EMISSION_TRANSVERSION, EMISSION_TRANSITION,
                           EMISSION_TRANSITION, EMISSION_TRANSVERSION, EMISSION_MATCH, EMISSION_TRANSVERSION,
                           EMISSION_TRANSVERSION, EMISSION_TRANSITION, EMISSION_TRANSVERSION, EMISSION_MATCH };
    memcpy(sM->EMISSION_MATCH_PROBS, M, sizeof M);
    for (int i = 0; i < 4; i++) sM->EMISSION_GAP_X_PROBS[i] = sM->EMISSION_GAP_Y_PROBS[i] = -1.6094379124341003;
}
StateMachine *stateMachine5_construct(StateMachineType type, int64_t parameterSetSize,
                                      void (*setEmissionsDefaults)(StateMachine *sM),
                                      double (*gapXProbFcn)(const double *, void *),
                                      double (*gapYProbFcn)(const double *, void *),
                                      double (*matchProbFcn)(const double *, void *, void *),
                                      void (*cellCalcUpdateExpFcn)(double *, double *, int64_t, int64_t, double,
                                                                   double, void *)) {
    if (type != fiveState && type != fiveStateAsymmetric) die("Wrong type for five state %i", (int) type);
    if (parameterSetSize != SYMBOL_NUMBER_NO_N)
        die("cpecan: the 5-state machine works on single bases (parameterSetSize 4)");
    StateMachine5 *s = calloc(1, sizeof *s);
    s->getXGapProbFcn = gapXProbFcn;
    s->getYGapProbFcn = gapYProbFcn;
    s->getMatchProbFcn = matchProbFcn;
    cpecan_sm5_set_functions(s);
    s->model.cellCalculateUpdateExpectations = cellCalcUpdateExpFcn;
    s->TRANSITION_MATCH_CONTINUE = -0.030064059121770816;
    s->TRANSITION_MATCH_FROM_SHORT_GAP_X = -1.272871422049609;
    s->TRANSITION_MATCH_FROM_LONG_GAP_X = -5.673280173170473;
    s->TRANSITION_GAP_SHORT_OPEN_X = -4.34381910900448;
    s->TRANSITION_GAP_SHORT_EXTEND_X = -0.3388262689231553;
    s->TRANSITION_GAP_SHORT_SWITCH_TO_X = -4.910694825551255;
    s->TRANSITION_GAP_LONG_OPEN_X = -6.30810595366929;
    s->TRANSITION_GAP_LONG_EXTEND_X = -0.003442492794189331;
    s->TRANSITION_GAP_LONG_SWITCH_TO_X = -6.30810595366929;
    s->TRANSITION_MATCH_FROM_SHORT_GAP_Y = s->TRANSITION_MATCH_FROM_SHORT_GAP_X;
    s->TRANSITION_MATCH_FROM_LONG_GAP_Y = s->TRANSITION_MATCH_FROM_LONG_GAP_X;
    s->TRANSITION_GAP_SHORT_OPEN_Y = s->TRANSITION_GAP_SHORT_OPEN_X;
    s->TRANSITION_GAP_SHORT_EXTEND_Y = s->TRANSITION_GAP_SHORT_EXTEND_X;
    s->TRANSITION_GAP_SHORT_SWITCH_TO_Y = s->TRANSITION_GAP_SHORT_SWITCH_TO_X;
    s->TRANSITION_GAP_LONG_OPEN_Y = s->TRANSITION_GAP_LONG_OPEN_X;
    s->TRANSITION_GAP_LONG_EXTEND_Y = s->TRANSITION_GAP_LONG_EXTEND_X;
    s->TRANSITION_GAP_LONG_SWITCH_TO_Y = s->TRANSITION_GAP_LONG_SWITCH_TO_X;
    s->model.type = type;
    s->model.parameterSetSize = parameterSetSize;
    s->model.stateNumber = 5;
    s->model.matchState = match;
    s->model.EMISSION_MATCH_PROBS = calloc(16, sizeof(double));
    s->model.EMISSION_GAP_X_PROBS = calloc(4, sizeof(double));
    s->model.EMISSION_GAP_Y_PROBS = calloc(4, sizeof(double));
    if (setEmissionsDefaults) setEmissionsDefaults((StateMachine *) s);
    return (StateMachine *) s;
}

static int read_doubles(FILE *f, double *dst, int64_t n) {
    for (int64_t i = 0; i < n; i++)
        if (fscanf(f, "%lf", &dst[i]) != 1) return 0;
    return 1;
}

StateMachine *getStrawManStateMachine3(const char *modelFile) {
    StateMachine3 *s = calloc(1, sizeof(StateMachine3));
    s->model.type = threeState;
    s->model.stateNumber = 3;
    s->model.matchState = match;
    s->model.parameterSetSize = NUM_OF_KMERS;
    const int64_t tableLen = 1 + NUM_OF_KMERS * MODEL_PARAMS;
    s->model.EMISSION_MATCH_PROBS = calloc((size_t) tableLen, sizeof(double));
    s->model.EMISSION_GAP_Y_PROBS = calloc((size_t) tableLen, sizeof(double));
    s->model.EMISSION_GAP_X_PROBS = calloc(NUM_OF_KMERS, sizeof(double));
    stateMachine3_setTransitionsToNanoporeDefaults((StateMachine *) s);
    s->getXGapProbFcn = emissions_kmer_getGapProb;
    s->getYGapProbFcn = emissions_signal_strawManGetKmerEventMatchProb;
    s->getMatchProbFcn = emissions_signal_strawManGetKmerEventMatchProb;
    cpecan_sm3_set_functions(s);
    s->model.cellCalculateUpdateExpectations = cell_signal_updateTransAndKmerSkipExpectations;
    for (int64_t i = 0; i < NUM_OF_KMERS; i++) s->model.EMISSION_GAP_X_PROBS[i] = -2.3025850929940455;
    if (modelFile) {
        /* 3 lines: match table, 30 skip bins (used by the vanilla/echelon models only), Y-gap table */
        FILE *f = fopen(modelFile, "r");
        double skip[30];
        if (!f) die("cpecan: cannot open pore model %s", modelFile);
        if (!read_doubles(f, s->model.EMISSION_MATCH_PROBS, tableLen) || !read_doubles(f, skip, 30) ||
            !read_doubles(f, s->model.EMISSION_GAP_Y_PROBS, tableLen))
            die("This stateMachine is not correct for signal model (%s)", modelFile);
        fclose(f);
    }
    return (StateMachine *) s;
}

/* impl/pairwiseAligner.c:1039-1063: the match runs of a forward-strand alignment as (x, y) pairs, `trim` positions cut
 * from either end of a run */
stList *convertPairwiseForwardStrandAlignmentToAnchorPairs(struct PairwiseAlignment *pA, int64_t trim) {
    stList *pairs = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    if (!pA->strand1 || !pA->strand2) die("cpecan: convertPairwiseForwardStrandAlignmentToAnchorPairs takes forward strands");
    int64_t j = pA->start1, k = pA->start2;
    for (int64_t i = 0; i < pA->operationList->length; i++) {
        const struct AlignmentOperation *op = pA->operationList->list[i];
        if (op->opType == PAIRWISE_MATCH)
            for (int64_t l = trim; l < op->length - trim; l++) stList_append(pairs, stIntTuple_construct2(j + l, k + l));
        if (op->opType != PAIRWISE_INDEL_Y) j += op->length;
        if (op->opType != PAIRWISE_INDEL_X) k += op->length;
    }
    if (j != pA->end1 || k != pA->end2) die("cpecan: the operations of the alignment do not span its intervals");
    return pairs;
}

StateMachine *getStateMachine4(const char *modelFile) { /* impl/stateMachine.c:1750-1759 */
    StateMachine *sM = stateMachine4_construct(fourState, NUM_OF_KMERS, emissions_signal_initEmissionsToZero,
                                               emissions_kmer_getGapProb, emissions_signal_strawManGetKmerEventMatchProb,
                                               emissions_signal_strawManGetKmerEventMatchProb,
                                               cell_signal_updateTransAndKmerSkipExpectations);
    if (modelFile) { /* emissions_signal_loadPoreModel :242-320: match table, 30 skip bins (not this machine's), Y-gap table */
        const int64_t tableLen = 1 + NUM_OF_KMERS * MODEL_PARAMS;
        FILE *f = fopen(modelFile, "r");
        double skip[30];
        if (!f) die("cpecan: cannot open pore model %s", modelFile);
        if (!read_doubles(f, sM->EMISSION_MATCH_PROBS, tableLen) || !read_doubles(f, skip, 30) ||
            !read_doubles(f, sM->EMISSION_GAP_Y_PROBS, tableLen))
            die("This stateMachine is not correct for signal model (%s)", modelFile);
        fclose(f);
    }
    return sM;
}

/* ---- NanoporeHDP: the reader of serialized HDPs (impl/nanopore_hdp.c:845-870, impl/hdp.c:3009-3273) ---- */
static char *read_line(FILE *f) { /* one line of any length, without the newline; NULL at EOF */
    size_t cap = 1 << 16, n = 0;
    char *buf = malloc(cap);
    int ch;
    while ((ch = fgetc(f)) != EOF && ch != '\n') {
        if (n + 2 > cap) buf = realloc(buf, cap *= 2);
        buf[n++] = (char) ch;
    }
    if (ch == EOF && n == 0) { free(buf); return NULL; }
    buf[n] = 0;
    return buf;
}
static int64_t line_int(FILE *f, const char *what) {
    char *l = read_line(f);
    if (!l) die("cpecan: truncated .nhdp (%s)", what);
    int64_t v = strtoll(l, NULL, 10);
    free(l);
    return v;
}
/* doubles of one line into dst (up to cap); returns how many the line holds */
static int64_t line_doubles(FILE *f, double *dst, int64_t cap) {
    char *l = read_line(f), *p = l, *e;
    int64_t n = 0;
    if (!l) die("cpecan: truncated .nhdp");
    for (;;) {
        double v = strtod(p, &e);
        if (e == p) break;
        if (n < cap && dst) dst[n] = v;
        n++;
        p = e;
    }
    free(l);
    return n;
}
NanoporeHDP *deserialize_nhdp(const char *filepath) {
    FILE *f = fopen(filepath, "r");
    if (!f) die("cpecan: cannot open %s", filepath);
    NanoporeHDP *h = calloc(1, sizeof *h);
    h->alphabetSize = line_int(f, "alphabet size");
    char *l = read_line(f);
    if (!l || h->alphabetSize < 1 || h->alphabetSize > 16 || (int64_t) strlen(l) < h->alphabetSize)
        die("cpecan: bad alphabet in %s", filepath);
    memcpy(h->alphabet, l, (size_t) h->alphabetSize);
    free(l);
    h->kmerLength = line_int(f, "k-mer length");
    if (h->kmerLength != KMER_LENGTH) die("cpecan: %s is not a 6-mer HDP", filepath);
    const int64_t splines = line_int(f, "splines finalized"), hasData = line_int(f, "has data");
    const int64_t sampleGamma = line_int(f, "sample gamma");
    h->numDps = line_int(f, "number of Dirichlet processes");
    if (!splines || !hasData) die("cpecan: %s does not hold a finalized HDP with data", filepath);
    (void) line_doubles(f, NULL, 0); /* data */
    (void) line_doubles(f, NULL, 0); /* Dirichlet process of each datum */
    (void) line_doubles(f, NULL, 0); /* base parameters mu nu alpha beta */
    double g[3];
    if (line_doubles(f, g, 3) != 3) die("cpecan: bad sampling grid in %s", filepath);
    h->gridLength = (int64_t) g[2];
    if (h->gridLength < 2 || !(g[0] < g[1])) die("cpecan: bad sampling grid in %s", filepath);
    h->grid = malloc(sizeof(double) * (size_t) h->gridLength);
    {   /* linspace (impl/hdp_math_utils.c:497-510) */
        const int64_t n = h->gridLength - 1;
        const double dx = (g[1] - g[0]) / ((double) n);
        for (int64_t i = 0; i < n; i++) h->grid[i] = g[0] + i * dx;
        h->grid[n] = g[1];
    }
    (void) line_doubles(f, NULL, 0); /* gamma */
    if (sampleGamma)
        for (int q = 0; q < 4; q++) (void) line_doubles(f, NULL, 0); /* gamma alpha, beta, w, s */
    int64_t *parent = malloc(sizeof(int64_t) * (size_t) h->numDps);
    for (int64_t d = 0; d < h->numDps; d++) {
        l = read_line(f);
        if (!l) die("cpecan: truncated .nhdp (parents)");
        parent[d] = l[0] == '-' ? -1 : strtoll(l, NULL, 10);
        free(l);
    }
    /* posterior predictives: a line per process, empty unless observed */
    int64_t *rowOf = malloc(sizeof(int64_t) * (size_t) h->numDps);
    size_t cap = 256;
    h->y = malloc(sizeof(double) * cap * (size_t) h->gridLength);
    for (int64_t d = 0; d < h->numDps; d++) {
        if ((size_t) h->nRows == cap) h->y = realloc(h->y, sizeof(double) * (cap *= 2) * (size_t) h->gridLength);
        const int64_t n = line_doubles(f, h->y + h->nRows * h->gridLength, h->gridLength);
        if (n != 0 && n != h->gridLength) die("cpecan: bad distribution for process %lld", (long long) d);
        rowOf[d] = n ? h->nRows++ : -1;
    }
    h->slope = malloc(sizeof(double) * (size_t) h->nRows * (size_t) h->gridLength);
    for (int64_t d = 0; d < h->numDps; d++) {
        double *dst = rowOf[d] >= 0 ? h->slope + rowOf[d] * h->gridLength : NULL;
        const int64_t n = line_doubles(f, dst, dst ? h->gridLength : 0);
        if ((n != 0) != (rowOf[d] >= 0) || (n != 0 && n != h->gridLength))
            die("cpecan: spline slopes of process %lld do not match its distribution", (long long) d);
    }
    fclose(f); /* the factor lines that follow are the sampler's state: not needed for densities */
    int64_t nK = 1;
    for (int q = 0; q < KMER_LENGTH; q++) nK *= h->alphabetSize;
    if (nK > h->numDps) die("cpecan: %s has fewer Dirichlet processes than k-mers", filepath);
    h->kmerRow = malloc(sizeof(int32_t) * (size_t) nK);
    for (int64_t k = 0; k < nK; k++) {
        int64_t d = k;
        while (d >= 0 && rowOf[d] < 0) d = parent[d];
        if (d < 0) die("cpecan: k-mer %lld has no observed ancestor", (long long) k);
        h->kmerRow[k] = (int32_t) rowOf[d];
    }
    free(parent);
    free(rowOf);
    return h;
}
void destroy_nanopore_hdp(NanoporeHDP *h) {
    if (!h) return;
    free(h->grid); free(h->y); free(h->slope); free(h->kmerRow); free(h);
}
int64_t get_nanopore_hdp_alphabet_size(NanoporeHDP *h) { return h->alphabetSize; }
char *get_nanopore_hdp_alphabet(NanoporeHDP *h) {
    char *c = calloc((size_t) h->alphabetSize + 1, 1);
    memcpy(c, h->alphabet, (size_t) h->alphabetSize);
    return c;
}
/* ---- NanoporeRead (impl/nanopore.c) ------------------------------------------------------------------ */
static void line_int64s(FILE *f, int64_t *dst, int64_t n, const char *what) {
    char *l = read_line(f), *p = l, *e;
    if (!l) die("cpecan: truncated .npRead (%s)", what);
    for (int64_t i = 0; i < n; i++) {
        dst[i] = strtoll(p, &e, 10);
        if (e == p) die("%s is not the correct length, should be %lld, got %lld", what, (long long) n, (long long) i);
        p = e;
    }
    (void) strtoll(p, &e, 10);
    if (e != p) die("%s is not the correct length, should be %lld", what, (long long) n);
    free(l);
}
NanoporeRead *nanopore_loadNanoporeReadFromFile(const char *nanoporeReadFile) {
    FILE *f = fopen(nanoporeReadFile, "r");
    if (!f) die("cpecan: cannot open %s", nanoporeReadFile);
    double hdr[13];
    if (line_doubles(f, hdr, 13) != 13) die("error parsing the header line of %s", nanoporeReadFile);
    NanoporeRead *r = calloc(1, sizeof *r);
    r->readLength = (int64_t) hdr[0];
    r->nbTemplateEvents = (int64_t) hdr[1];
    r->nbComplementEvents = (int64_t) hdr[2];
    const NanoporeReadAdjustmentParameters t = { hdr[3], hdr[4], hdr[5], hdr[6], hdr[7] };
    const NanoporeReadAdjustmentParameters c = { hdr[8], hdr[9], hdr[10], hdr[11], hdr[12] };
    r->templateParams = t;
    r->complementParams = c;
    r->twoDread = read_line(f);
    if (!r->twoDread) die("error parsing read from npRead file");
    for (char *q = r->twoDread; *q; q++) /* the reference reads it with %s: first token only */
        if (*q == ' ' || *q == '\t' || *q == '\r') { *q = 0; break; }
    r->templateEventMap = malloc(sizeof(int64_t) * (size_t) (r->readLength + 1));
    line_int64s(f, r->templateEventMap, r->readLength, "template event map");
    r->templateEvents = malloc(sizeof(double) * (size_t) (r->nbTemplateEvents * NB_EVENT_PARAMS + 1));
    if (line_doubles(f, r->templateEvents, r->nbTemplateEvents * NB_EVENT_PARAMS) != r->nbTemplateEvents * NB_EVENT_PARAMS)
        die("incorrect number of template events, should be %lld", (long long) r->nbTemplateEvents);
    r->complementEventMap = malloc(sizeof(int64_t) * (size_t) (r->readLength + 1));
    line_int64s(f, r->complementEventMap, r->readLength, "complement event map");
    r->complementEvents = malloc(sizeof(double) * (size_t) (r->nbComplementEvents * NB_EVENT_PARAMS + 1));
    if (line_doubles(f, r->complementEvents, r->nbComplementEvents * NB_EVENT_PARAMS) !=
        r->nbComplementEvents * NB_EVENT_PARAMS)
        die("incorrect number of complement events, should be %lld", (long long) r->nbComplementEvents);
    r->scaled = true;
    fclose(f);
    return r;
}
stList *nanopore_remapAnchorPairs(stList *anchorPairs, int64_t *eventMap) {
    return nanopore_remapAnchorPairsWithOffset(anchorPairs, eventMap, -1);
}
/* (x, y) in 2D-read coordinates -> (x, event index); mapOffset >= 0 re-bases on that position's event */
stList *nanopore_remapAnchorPairsWithOffset(stList *unmappedPairs, int64_t *eventMap, int64_t mapOffset) {
    stList *mapped = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    const int64_t base = mapOffset >= 0 ? eventMap[mapOffset] : 0;
    for (int64_t i = 0; i < stList_length(unmappedPairs); i++) {
        stIntTuple *p = stList_get(unmappedPairs, i);
        stList_append(mapped, stIntTuple_construct2(stIntTuple_get(p, 0), eventMap[stIntTuple_get(p, 1)] - base));
    }
    return mapped;
}
void nanopore_descaleNanoporeRead(NanoporeRead *r) {
    /* nanopore_descaleEvents (:34-38) walks i < nb_events in steps of NB_EVENT_PARAMS over the flat array,
     * so only the means of the first third of the events are de-scaled: kept as is */
    for (int64_t i = 0; i < r->nbTemplateEvents; i += NB_EVENT_PARAMS)
        r->templateEvents[i] = (r->templateEvents[i] - r->templateParams.shift) / r->templateParams.scale;
    for (int64_t i = 0; i < r->nbComplementEvents; i += NB_EVENT_PARAMS)
        r->complementEvents[i] = (r->complementEvents[i] - r->complementParams.shift) / r->complementParams.scale;
    r->scaled = false;
}
void nanopore_nanoporeReadDestruct(NanoporeRead *r) {
    free(r->twoDread); free(r->templateEventMap); free(r->templateEvents);
    free(r->complementEventMap); free(r->complementEvents); free(r);
}

/* the C-ABI's view of an HDP state machine (cpecan_hdp_model, cpecan_hip.h): pointers into the NanoporeHDP, no copy */
void cpecan_hdp_machine_as_model(StateMachine *sM, void *out) {
    if (sM->type != threeStateHdp || !((StateMachine3_HDP *) sM)->hdpModel)
        die("cpecan_hdp_machine_as_model takes a StateMachine3_HDP with its NanoporeHDP");
    const StateMachine3_HDP *sh = (const StateMachine3_HDP *) sM;
    const NanoporeHDP *nh = sh->hdpModel;
    cpecan_hdp_model *m = out;
    memcpy(m->transitions, &sh->TRANSITION_MATCH_CONTINUE, sizeof m->transitions);
    m->alphabet = nh->alphabet;
    m->alphabet_size = (int32_t) nh->alphabetSize;
    m->grid_length = (int32_t) nh->gridLength;
    m->grid = nh->grid;
    m->n_rows = nh->nRows;
    m->posterior_predictive = nh->y;
    m->spline_slopes = nh->slope;
    m->kmer_row = nh->kmerRow;
}

StateMachine *getHdpStateMachine3(NanoporeHDP *hdp) { /* impl/stateMachine.c:1738-1748 */
    /* through the plug-in constructor, as the reference does: the (zeroed) emission tables exist -- vanillaAlign hands
     * EMISSION_MATCH_PROBS of this machine to writePosteriorProbs -- and the k-mer gap table holds log(0.1) */
    return stateMachine3Hdp_construct(threeStateHdp, NUM_OF_KMERS, stateMachine3_setTransitionsToNanoporeDefaults,
                                      emissions_signal_initEmissionsToZero, hdp, emissions_kmer_getGapProb,
                                      get_nanopore_kmer_density, get_nanopore_kmer_density,
                                      cell_signal_updateTransAndKmerSkipExpectations2);
}

StateMachine *getSignalStateMachine3Vanilla(const char *modelFile) {
    StateMachine3Vanilla *s = calloc(1, sizeof(StateMachine3Vanilla));
    s->model.type = vanilla;
    s->model.stateNumber = 3;
    s->model.matchState = match;
    s->model.parameterSetSize = NUM_OF_KMERS;
    s->TRANSITION_M_TO_Y_NOT_X = 0.17;
    s->TRANSITION_E_TO_E = 0.55f;
    s->DEFAULT_END_MATCH_PROB = -0.23552123624314988;
    s->DEFAULT_END_FROM_X_PROB = -1.6269694202638481;
    s->DEFAULT_END_FROM_Y_PROB = -4.3187242127300092;
    s->getKmerSkipProb = emissions_signal_getBetaOrAlphaSkipProb;
    s->getScaledMatchProbFcn = emissions_signal_getEventMatchProbWithTwoDists;
    s->getMatchProbFcn = emissions_signal_getEventMatchProbWithTwoDists;
    cpecan_sm3vanilla_set_functions(s);
    s->model.cellCalculateUpdateExpectations = cell_signal_updateBetaAndAlphaProb;
    const int64_t tableLen = 1 + NUM_OF_KMERS * MODEL_PARAMS;
    s->model.EMISSION_MATCH_PROBS = calloc((size_t) tableLen, sizeof(double));
    s->model.EMISSION_GAP_Y_PROBS = calloc((size_t) tableLen, sizeof(double));
    s->model.EMISSION_GAP_X_PROBS = calloc(60, sizeof(double)); /* skip bins: beta[30] | alpha[30] */
    if (modelFile) {
        FILE *f = fopen(modelFile, "r");
        if (!f) die("cpecan: cannot open pore model %s", modelFile);
        if (!read_doubles(f, s->model.EMISSION_MATCH_PROBS, tableLen) ||
            !read_doubles(f, s->model.EMISSION_GAP_X_PROBS, 30) ||
            !read_doubles(f, s->model.EMISSION_GAP_Y_PROBS, tableLen))
            die("This stateMachine is not correct for signal model (%s)", modelFile);
        fclose(f);
        for (int i = 0; i < 30; i++) s->model.EMISSION_GAP_X_PROBS[i + 30] = s->model.EMISSION_GAP_X_PROBS[i];
    }
    return (StateMachine *) s;
}
void stateMachine3Vanilla_setStrandTransitionsToDefaults(StateMachine *sM, Strand strand) {
    StateMachine3Vanilla *s = (StateMachine3Vanilla *) sM;
    s->TRANSITION_M_TO_Y_NOT_X = strand == template ? 0.17f : 0.14f;
    s->TRANSITION_E_TO_E = strand == template ? 0.55f : 0.49f;
}

void emissions_signal_scaleModel(StateMachine *sM, double scale, double shift, double var,
                                 double scale_sd, double var_sd) {
    double *m = sM->EMISSION_MATCH_PROBS;
    for (int64_t i = 1; i < (sM->parameterSetSize * MODEL_PARAMS) + 1; i += MODEL_PARAMS) {
        m[i] = m[i] * scale + shift;
        m[i + 1] = m[i + 1] * var;
        m[i + 2] = m[i + 2] * scale_sd;
        m[i + 4] = m[i + 4] * var_sd;
        m[i + 3] = sqrt(pow(m[i + 2], 3.0) / m[i + 4]);
    }
}

int64_t emissions_discrete_getKmerIndex(void *kmer) {
    const char *k = kmer;
    int64_t x = 0, l = NUM_OF_KMERS / 4;
    for (int i = 0; i < KMER_LENGTH; i++) {
        int64_t b = k[i] == 'A' ? 0 : k[i] == 'C' ? 1 : k[i] == 'G' ? 2 : k[i] == 'T' ? 3 : NUM_OF_KMERS + 1;
        x += (i < KMER_LENGTH - 1 ? l : 1) * b;
        l /= 4;
    }
    return x;
}

int64_t emissions_discrete_getKmerIndexFromKmer(void *kmer) { return emissions_discrete_getKmerIndex(kmer); }

void stateMachine_destruct(StateMachine *sM) {
    if (!sM) return;
    free(sM->EMISSION_MATCH_PROBS);
    free(sM->EMISSION_GAP_X_PROBS);
    free(sM->EMISSION_GAP_Y_PROBS);
    free(sM);
}

/* ---- Diagonal / Band / BandIterator / logAdd / filterToRemoveOverlap (host integer utilities) ---------- */
Diagonal diagonal_construct(int64_t xay, int64_t xmyL, int64_t xmyR) {
    if ((xay + xmyL) % 2 != 0 || (xay + xmyR) % 2 != 0 || xmyL > xmyR)
        die("PAIRWISE_ALIGNMENT_EXCEPTION: Attempt to create diagonal with invalid coordinates: xay %lld xmyL %lld "
            "xmyR %lld", (long long) xay, (long long) xmyL, (long long) xmyR);
    Diagonal d = { xay, xmyL, xmyR };
    return d;
}
int64_t diagonal_getXay(Diagonal d) { return d.xay; }
int64_t diagonal_getMinXmy(Diagonal d) { return d.xmyL; }
int64_t diagonal_getMaxXmy(Diagonal d) { return d.xmyR; }
int64_t diagonal_getWidth(Diagonal d) { return (d.xmyR - d.xmyL) / 2 + 1; }
int64_t diagonal_getXCoordinate(int64_t xay, int64_t xmy) { return (xay + xmy) / 2; }
int64_t diagonal_getYCoordinate(int64_t xay, int64_t xmy) { return (xay - xmy) / 2; }
int64_t diagonal_equals(Diagonal a, Diagonal b) { return a.xay == b.xay && a.xmyL == b.xmyL && a.xmyR == b.xmyR; }

struct _band {
    Diagonal *diagonals;
    int64_t lXalY;
};
Band *band_construct(stList *anchorPairs, int64_t lX, int64_t lY, int64_t expansion) {
    const int64_t na = anchorPairs ? stList_length(anchorPairs) : 0, n = lX + lY + 1;
    int64_t *a = malloc(sizeof(int64_t) * 2 * (size_t) (na + 1));
    for (int64_t k = 0; k < na; k++) {
        a[2 * k] = stIntTuple_get(stList_get(anchorPairs, k), 0);
        a[2 * k + 1] = stIntTuple_get(stList_get(anchorPairs, k), 1);
    }
    int32_t *L = malloc(sizeof(int32_t) * (size_t) n), *R = malloc(sizeof(int32_t) * (size_t) n);
    CHECK(cpecan_band_construct(a, na, lX, lY, expansion, L, R));
    Band *b = malloc(sizeof *b);
    b->lXalY = lX + lY;
    b->diagonals = malloc(sizeof(Diagonal) * (size_t) n);
    for (int64_t d = 0; d < n; d++) b->diagonals[d] = diagonal_construct(d, L[d], R[d]);
    free(a); free(L); free(R);
    return b;
}
void band_destruct(Band *band) {
    free(band->diagonals);
    free(band);
}
struct _bandIterator {
    Band *band;
    int64_t index;
};
BandIterator *bandIterator_construct(Band *band) {
    BandIterator *it = malloc(sizeof *it);
    it->band = band;
    it->index = 0;
    return it;
}
BandIterator *bandIterator_clone(BandIterator *it) {
    BandIterator *c = malloc(sizeof *c);
    *c = *it;
    return c;
}
void bandIterator_destruct(BandIterator *it) { free(it); }
/* past either end the iterator keeps returning the end diagonal (:213-227) */
Diagonal bandIterator_getNext(BandIterator *it) {
    const int64_t last = it->band->lXalY;
    const Diagonal d = it->band->diagonals[it->index > last ? last : it->index];
    if (it->index <= last) it->index++;
    return d;
}
Diagonal bandIterator_getPrevious(BandIterator *it) {
    if (it->index > 0) it->index--;
    return it->band->diagonals[it->index];
}

/* lookup() / logAdd() :238-255: log(exp(x) + exp(y)) by a piecewise cubic of the gap, with float-suffixed
 * coefficients promoted to double, the larger operand returned as is beyond a gap of 7.5 */
static double logadd_lookup(double x) {
    if (x <= 2.50f) {
        if (x <= 1.00f)
            return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x + 0.693203116424741f;
        return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x + 0.692140569840976f;
    }
    if (x <= 4.50f)
        return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x + 0.514272634594009f;
    return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x + 0.168037164329057f;
}
double logAdd(double x, double y) {
    if (x < y) return (x == LOG_ZERO || y - x >= 7.5) ? y : logadd_lookup(y - x) + x;
    return (y == LOG_ZERO || x - y >= 7.5) ? x : logadd_lookup(x - y) + y;
}

stList *filterToRemoveOverlap(stList *pairs) {
    const int64_t n = stList_length(pairs);
    int64_t *x = malloc(sizeof(int64_t) * (size_t) (n + 1)), *y = malloc(sizeof(int64_t) * (size_t) (n + 1));
    char *below = calloc((size_t) n + 1, 1); /* smaller in both coordinates than everything after it */
    for (int64_t i = 0; i < n; i++) {
        x[i] = stIntTuple_get(stList_get(pairs, i), 0);
        y[i] = stIntTuple_get(stList_get(pairs, i), 1);
    }
    int64_t mx = INT64_MAX, my = INT64_MAX;
    for (int64_t i = n - 1; i >= 0; i--) {
        below[i] = x[i] < mx && y[i] < my;
        if (x[i] < mx) mx = x[i];
        if (y[i] < my) my = y[i];
    }
    /* the reference looks candidates up by VALUE: an equal pair earlier in the list counts as marked too */
    for (int64_t i = n - 2; i >= 0; i--)
        if (!below[i] && below[i + 1] && x[i] == x[i + 1] && y[i] == y[i + 1]) below[i] = 1;
    stList *out = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    mx = INT64_MIN;
    my = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        if (x[i] > mx && y[i] > my && below[i]) stList_append(out, stIntTuple_construct2(x[i], y[i]));
        if (x[i] > mx) mx = x[i];
        if (y[i] > my) my = y[i];
    }
    free(x); free(y); free(below);
    return out;
}

stList *getSplitPoints(stList *anchorPairs, int64_t lX, int64_t lY, int64_t maxMatrixSize,
                       bool raggedL, bool raggedR) {
    int64_t n = stList_length(anchorPairs);
    int64_t *a = malloc(sizeof(int64_t) * 2 * (size_t) (n + 1));
    for (int64_t i = 0; i < n; i++) {
        a[2 * i] = stIntTuple_get(stList_get(anchorPairs, i), 0);
        a[2 * i + 1] = stIntTuple_get(stList_get(anchorPairs, i), 1);
    }
    int64_t *out = malloc(sizeof(int64_t) * 4 * (size_t) (n + 2));
    int64_t m = cpecan_split_points(a, n, lX, lY, maxMatrixSize, raggedL, raggedR, out, n + 2);
    stList *l = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    for (int64_t i = 0; i < m; i++) stList_append(l, tuple4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]));
    free(a);
    free(out);
    return l;
}

/* ------------------------------------------------------------------------------------------------ */
/* the GPU context of this process                                                                  */
/* ------------------------------------------------------------------------------------------------ */
/* One context (device tables, stream) per calling thread: the reference's callers align the template and the
 * complement strand in two OpenMP sections (vanillaAlign.c:737-800), and the two then run side by side on the GPU
 * instead of queueing behind a process-wide lock.  A thread's context goes with the thread. */
static pthread_key_t g_ctx_key;
static pthread_once_t g_ctx_once = PTHREAD_ONCE_INIT;
static void ctx_release(void *ctx) {
    if (ctx) cpecan_hip_ctx_destroy((cpecan_ctx *) ctx);
}
static void ctx_key_init(void) { pthread_key_create(&g_ctx_key, ctx_release); }

static cpecan_ctx *context(void) {
    pthread_once(&g_ctx_once, ctx_key_init);
    cpecan_ctx *ctx = pthread_getspecific(g_ctx_key);
    if (!ctx) {
        const char *dev = getenv("CPECAN_DEVICE");
        CHECK(cpecan_hip_ctx_create(dev ? atoi(dev) : 0, &ctx));
        pthread_setspecific(g_ctx_key, ctx);
    }
    return ctx;
}

/* returns 1 for the DNA-against-DNA combination (5-state machine, sequence_getBase on both sides),
 * 2 for k-mers against events under the vanilla machine (sequence_getKmer2), 3 under the HDP machine
 * (sequence_getKmer3), 4 under the 4-state signal machine and 0 under the 3-state strawMan machine (both
 * sequence_getKmer); anything else is not on the GPU path */
static int check_known_combination(StateMachine *sM, Sequence *sX, Sequence *sY) {
    if (!cpecan_sm_functions_known(sM))
        die("cpecan: this StateMachine carries a cellCalculate or emission function of the caller's own; the GPU path "
            "implements the reference's models only (symbol, strawMan, vanilla two-distribution and HDP emissions) "
            "and there is no CPU path to fall back to");
    if ((sM->type == fiveState || sM->type == fiveStateAsymmetric) && sM->stateNumber == 5) {
        if (sX->get != sequence_getBase || sY->get != sequence_getBase)
            die("cpecan: the 5-state machine needs sequence_getBase element getters on both sequences");
        return 1;
    }
    if (sM->type == threeStateHdp && sM->stateNumber == 3) {
        if (sX->get != sequence_getKmer3 || sY->get != sequence_getEvent)
            die("cpecan: the HDP machine needs sequence_getKmer3 / sequence_getEvent element getters");
        if (!((StateMachine3_HDP *) sM)->hdpModel) die("cpecan: the HDP machine has no NanoporeHDP");
        return 3;
    }
    if (sM->type == vanilla && sM->stateNumber == 3) {
        if (sX->get != sequence_getKmer2 || sY->get != sequence_getEvent)
            die("cpecan: the vanilla machine needs sequence_getKmer2 / sequence_getEvent element getters");
        return 2;
    }
    if (sM->type == fourState && sM->stateNumber == 4) {
        if (sX->get != sequence_getKmer || sY->get != sequence_getEvent)
            die("cpecan: the 4-state machine needs sequence_getKmer / sequence_getEvent element getters");
        return 4;
    }
    if (sM->type != threeState || sM->stateNumber != 3)
        die("cpecan: only the threeState (strawMan), fourState, vanilla and HDP signal StateMachines and the fiveState "
            "symbol StateMachine run on the GPU path (type %d)", sM->type);
    if (sX->get != sequence_getKmer || sY->get != sequence_getEvent)
        die("cpecan: the GPU path needs sequence_getKmer / sequence_getEvent element getters");
    return 0;
}

typedef struct {
    int64_t read, x1, y1;
} ItemOrigin;

/* Shared driver: n reads -> one batch.  mode 0: aligned pairs into lists[i]; mode 1: expectations
 * added into hmm (all reads must then share sMs[0]). */
static void run_reads(int64_t n, StateMachine **sMs, Sequence **sXs, Sequence **sYs, stList **anchorLists,
                      PairwiseAlignmentParameters *p, bool raggedL, bool raggedR, int mode, int unbanded,
                      stList **lists, void *hmmOut) {
    cpecan_ctx *ctx = context();
    int64_t nX = 0, nY = 0, nA = 0, nItems = 0, capItems = 0;
    const int kind = n > 0 ? check_known_combination(sMs[0], sXs[0], sYs[0]) : 0;
    const int dna = kind == 1, van = kind == 2, hdp = kind == 3, sm4 = kind == 4;
    if (sm4 && mode != 0) die("cpecan: the 4-state machine has no expectations (the reference has no Hmm for it)");
    const int64_t xPad = dna ? 0 : KMER_LENGTH - 1; /* a k-mer sequence of lX elements spans lX + 5 chars */
    for (int64_t i = 0; i < n; i++) {
        if (check_known_combination(sMs[i], sXs[i], sYs[i]) != kind)
            die("cpecan: one call cannot mix state machine kinds");
        nX += sXs[i]->length + (sXs[i]->length > 0 ? xPad : 0);
        nY += sYs[i]->length;
        nA += anchorLists && anchorLists[i] ? stList_length(anchorLists[i]) : 0;
    }
    char *chars = malloc((size_t) nX + 8);
    double *events = malloc(dna ? 8 : sizeof(double) * 3 * (size_t) (nY + 1));
    char *ychars = malloc(dna ? (size_t) nY + 8 : 8);
    int64_t *anchors = malloc(sizeof(int64_t) * 2 * (size_t) (nA + 1));
    cpecan_item *items = NULL;
    ItemOrigin *origin = NULL;
    int64_t *firstItem = malloc(sizeof(int64_t) * (size_t) (n + 1));

    /* models: one per distinct StateMachine pointer */
    cpecan_sm3_model *models = malloc(sizeof(cpecan_sm3_model) * (size_t) n);
    int32_t *modelOf = malloc(sizeof(int32_t) * (size_t) n);
    int32_t nModels = 0;
    cpecan_sm5_model *models5 = malloc(sizeof(cpecan_sm5_model) * (size_t) (dna ? n : 1));
    StateMachine **owner5 = malloc(sizeof(StateMachine *) * (size_t) (n + 1));
    for (int64_t i = 0; dna && i < n; i++) { /* one model per distinct StateMachine5 */
        int32_t found = -1;
        for (int32_t k = 0; k < nModels && found < 0; k++)
            if (owner5[k] == sMs[i]) found = k;
        if (found < 0) {
            const StateMachine5 *s5 = (const StateMachine5 *) sMs[i];
            cpecan_sm5_model *m = &models5[nModels];
            memcpy(m->transitions, &s5->TRANSITION_MATCH_CONTINUE, sizeof m->transitions);
            memcpy(m->match_probs, sMs[i]->EMISSION_MATCH_PROBS, sizeof m->match_probs);
            memcpy(m->gap_x_probs, sMs[i]->EMISSION_GAP_X_PROBS, sizeof m->gap_x_probs);
            memcpy(m->gap_y_probs, sMs[i]->EMISSION_GAP_Y_PROBS, sizeof m->gap_y_probs);
            owner5[nModels] = sMs[i];
            found = nModels++;
        }
        modelOf[i] = found;
    }
    cpecan_vanilla_model *modelsV = malloc(sizeof(cpecan_vanilla_model) * (size_t) (van ? n : 1));
    for (int64_t i = 0; van && i < n; i++) { /* one model per distinct StateMachine3Vanilla */
        int32_t found = -1;
        for (int32_t k = 0; k < nModels && found < 0; k++)
            if (owner5[k] == sMs[i]) found = k;
        if (found < 0) {
            const StateMachine3Vanilla *sv = (const StateMachine3Vanilla *) sMs[i];
            cpecan_vanilla_model *m = &modelsV[nModels];
            m->m_to_y_not_x = sv->TRANSITION_M_TO_Y_NOT_X;
            m->e_to_e = sv->TRANSITION_E_TO_E;
            m->end_match_prob = sv->DEFAULT_END_MATCH_PROB;
            m->end_from_x_prob = sv->DEFAULT_END_FROM_X_PROB;
            m->end_from_y_prob = sv->DEFAULT_END_FROM_Y_PROB;
            m->match_probs = sMs[i]->EMISSION_MATCH_PROBS;
            m->skip_probs = sMs[i]->EMISSION_GAP_X_PROBS;
            m->gap_y_probs = sMs[i]->EMISSION_GAP_Y_PROBS;
            owner5[nModels] = sMs[i];
            found = nModels++;
        }
        modelOf[i] = found;
    }
    cpecan_hdp_model *modelsH = malloc(sizeof(cpecan_hdp_model) * (size_t) (hdp ? n : 1));
    for (int64_t i = 0; hdp && i < n; i++) { /* one model per distinct StateMachine3_HDP */
        int32_t found = -1;
        for (int32_t k = 0; k < nModels && found < 0; k++)
            if (owner5[k] == sMs[i]) found = k;
        if (found < 0) {
            cpecan_hdp_machine_as_model(sMs[i], &modelsH[nModels]);
            owner5[nModels] = sMs[i];
            found = nModels++;
        }
        modelOf[i] = found;
    }
    for (int64_t i = 0; kind == 0 && i < n; i++) {
        StateMachine3 *s3 = (StateMachine3 *) sMs[i];
        const double t[9] = { s3->TRANSITION_MATCH_CONTINUE, s3->TRANSITION_MATCH_FROM_GAP_X,
                              s3->TRANSITION_MATCH_FROM_GAP_Y, s3->TRANSITION_GAP_OPEN_X,
                              s3->TRANSITION_GAP_OPEN_Y, s3->TRANSITION_GAP_EXTEND_X,
                              s3->TRANSITION_GAP_EXTEND_Y, s3->TRANSITION_GAP_SWITCH_TO_X,
                              s3->TRANSITION_GAP_SWITCH_TO_Y };
        int32_t found = -1;
        for (int32_t k = 0; k < nModels && found < 0; k++)
            if (models[k].match_probs == sMs[i]->EMISSION_MATCH_PROBS &&
                models[k].gap_x_probs == sMs[i]->EMISSION_GAP_X_PROBS &&
                models[k].gap_y_probs == sMs[i]->EMISSION_GAP_Y_PROBS &&
                memcmp(models[k].transitions, t, sizeof t) == 0)
                found = k;
        if (found < 0) {
            cpecan_sm3_model *m = &models[nModels];
            memcpy(m->transitions, t, sizeof t);
            m->match_probs = sMs[i]->EMISSION_MATCH_PROBS;
            m->gap_x_probs = sMs[i]->EMISSION_GAP_X_PROBS;
            m->gap_y_probs = sMs[i]->EMISSION_GAP_Y_PROBS;
            found = nModels++;
        }
        modelOf[i] = found;
    }
    cpecan_sm4_model *models4 = malloc(sizeof(cpecan_sm4_model) * (size_t) (sm4 ? n : 1));
    for (int64_t i = 0; sm4 && i < n; i++) { /* one model per distinct StateMachine4 */
        int32_t found = -1;
        for (int32_t k = 0; k < nModels && found < 0; k++)
            if (owner5[k] == sMs[i]) found = k;
        if (found < 0) {
            const StateMachine4 *s4 = (const StateMachine4 *) sMs[i];
            cpecan_sm4_model *m = &models4[nModels];
            memcpy(m->transitions, &s4->TRANSITION_MATCH_CONTINUE, sizeof m->transitions);
            m->match_probs = sMs[i]->EMISSION_MATCH_PROBS;
            m->gap_x_probs = sMs[i]->EMISSION_GAP_X_PROBS;
            m->gap_y_probs = sMs[i]->EMISSION_GAP_Y_PROBS;
            owner5[nModels] = sMs[i];
            found = nModels++;
        }
        modelOf[i] = found;
    }
    int32_t *ids = malloc(sizeof(int32_t) * (size_t) nModels);
    CHECK(cpecan_hip_models_clear(ctx));
    if (dna) CHECK(cpecan_hip_models5_create(ctx, models5, nModels, ids));
    else if (van) CHECK(cpecan_hip_modelsv_create(ctx, modelsV, nModels, 0, ids));
    else if (hdp) CHECK(cpecan_hip_modelsh_create(ctx, modelsH, nModels, ids));
    else if (sm4) CHECK(cpecan_hip_models4_create(ctx, models4, nModels, ids));
    else CHECK(cpecan_hip_models_create(ctx, models, nModels, 0, ids));

    int64_t xo = 0, yo = 0, ao = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t lX = sXs[i]->length, lY = sYs[i]->length;
        const int64_t na = anchorLists && anchorLists[i] ? stList_length(anchorLists[i]) : 0;
        if (lX > 0) memcpy(chars + xo, sXs[i]->elements, (size_t) (lX + xPad));
        if (lY > 0 && dna) memcpy(ychars + yo, sYs[i]->elements, (size_t) lY);
        if (lY > 0 && !dna) memcpy(events + 3 * yo, sYs[i]->elements, sizeof(double) * 3 * (size_t) lY);
        int64_t *ra = malloc(sizeof(int64_t) * 2 * (size_t) (na + 1));
        for (int64_t k = 0; k < na; k++) {
            ra[2 * k] = stIntTuple_get(stList_get(anchorLists[i], k), 0);
            ra[2 * k + 1] = stIntTuple_get(stList_get(anchorLists[i], k), 1);
        }
        /* getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps :1356-1422 */
        int64_t *sp = malloc(sizeof(int64_t) * 4 * (size_t) (na + 2));
        int64_t nSp;
        if (unbanded) { /* 1: getAlignedPairsWithoutBanding; 2: one getPosteriorProbsWithBanding call */
            nSp = 1; sp[0] = 0; sp[1] = 0; sp[2] = lX; sp[3] = lY;
        } else {
            nSp = cpecan_split_points(ra, na, lX, lY, p->splitMatrixBiggerThanThis, raggedL, raggedR, sp, na + 2);
        }
        firstItem[i] = nItems;
        int64_t j = 0;
        for (int64_t r = 0; r < nSp; r++) {
            const int64_t x1 = sp[4 * r], y1 = sp[4 * r + 1], x2 = sp[4 * r + 2], y2 = sp[4 * r + 3];
            if (nItems == capItems) {
                capItems = capItems ? capItems * 2 : 64;
                items = realloc(items, sizeof(cpecan_item) * (size_t) capItems);
                origin = realloc(origin, sizeof(ItemOrigin) * (size_t) capItems);
            }
            cpecan_item *it = &items[nItems];
            memset(it, 0, sizeof *it);
            it->x_offset = xo + x1; it->lX = x2 - x1;
            it->y_offset = yo + y1; it->lY = y2 - y1;
            it->anchor_offset = ao;
            while (j < na && ra[2 * j] + ra[2 * j + 1] < x2 + y2) {
                anchors[2 * ao] = ra[2 * j] - x1;
                anchors[2 * ao + 1] = ra[2 * j + 1] - y1;
                ao++; j++;
            }
            it->n_anchors = ao - it->anchor_offset;
            it->model_id = ids[modelOf[i]];
            it->ragged_left = raggedL || r > 0;
            it->ragged_right = raggedR || r < nSp - 1;
            origin[nItems].read = i; origin[nItems].x1 = x1; origin[nItems].y1 = y1;
            nItems++;
        }
        xo += lX + (lX > 0 ? xPad : 0);
        yo += lY;
        free(ra);
        free(sp);
    }
    firstItem[n] = nItems;

    if (nItems > 0) {
        cpecan_band_params bp = { p->threshold, p->minDiagsBetweenTraceBack, p->traceBackDiagonals,
                                  p->diagonalExpansion };
        if (hdp && mode) bp.threshold = ((HdpHmmExpectations *) hmmOut)->threshold; /* the assignment bar */
        cpecan_batch *batch = NULL;
        if (dna)
            CHECK(cpecan_hip_batch_create_dna(ctx, items, nItems, chars, xo, ychars, yo, anchors, ao, &bp,
                                              (unbanded == 1 ? CPECAN_FLAG_UNBANDED : 0) |
                                                  (mode ? CPECAN_FLAG_EXPECTATIONS : 0), &batch));
        else if (hdp)
            CHECK(cpecan_hip_batch_create_hdp(ctx, items, nItems, chars, xo, events, yo, anchors, ao, &bp,
                                              (unbanded == 1 ? CPECAN_FLAG_UNBANDED : 0) |
                                                  (mode ? CPECAN_FLAG_EXPECTATIONS : 0), &batch));
        else if (sm4)
            CHECK(cpecan_hip_batch_create_sm4(ctx, items, nItems, chars, xo, events, yo, anchors, ao, &bp,
                                              unbanded == 1 ? CPECAN_FLAG_UNBANDED : 0, &batch));
        else if (van)
            CHECK(cpecan_hip_batch_create_vanilla(ctx, items, nItems, chars, xo, events, yo, anchors, ao, &bp,
                                                  (unbanded == 1 ? CPECAN_FLAG_UNBANDED : 0) |
                                                      (mode ? CPECAN_FLAG_EXPECTATIONS : 0), &batch));
        else
            CHECK(cpecan_hip_batch_create(ctx, items, nItems, chars, xo, events, yo, anchors, ao, &bp,
                                          mode ? CPECAN_MODE_EXPECTATIONS : CPECAN_MODE_POSTERIOR,
                                          CPECAN_KERNEL_AUTO,
                                          /* (one batch per call, nothing chained behind it: the smaller footprint) */
                                          (unbanded == 1 ? CPECAN_FLAG_UNBANDED : 0) | CPECAN_FLAG_SMALL_FOOTPRINT, &batch));
        CHECK(cpecan_hip_batch_run(batch));
        CHECK(cpecan_hip_batch_sync(batch));
        if (mode == 0) {
            int64_t *np = malloc(sizeof(int64_t) * (size_t) nItems);
            CHECK(cpecan_hip_batch_counts(batch, np, NULL, NULL));
            for (int64_t i = 0; i < n; i++) {
                lists[i] = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
                for (int64_t k = firstItem[i]; k < firstItem[i + 1]; k++) {
                    int64_t *tri = malloc(sizeof(int64_t) * 3 * (size_t) (np[k] + 1));
                    CHECK(cpecan_hip_batch_fetch_pairs(batch, k, tri, NULL, np[k] + 1));
                    if (unbanded == 2) { /* the order diagonalCalculationPosteriorMatchProbs appends in */
                        for (int64_t q = 0; q < np[k]; q++)
                            stList_append(lists[i], stIntTuple_construct3(tri[3 * q], tri[3 * q + 1], tri[3 * q + 2]));
                    } else if (unbanded) {
                        /* getAlignedPairsWithoutBanding walks the diagonals upwards (:1560): groups of
                         * equal x+y in reverse group order, order inside a group kept */
                        int64_t e = np[k];
                        while (e > 0) {
                            int64_t s = e - 1;
                            const int64_t d = tri[3 * s + 1] + tri[3 * s + 2];
                            while (s > 0 && tri[3 * (s - 1) + 1] + tri[3 * (s - 1) + 2] == d) s--;
                            for (int64_t q = s; q < e; q++)
                                stList_append(lists[i], stIntTuple_construct3(tri[3 * q], tri[3 * q + 1], tri[3 * q + 2]));
                            e = s;
                        }
                    } else {
                        /* shift back to the read's coordinates, then tail first (stList_pop, :1451-1453) */
                        for (int64_t q = np[k] - 1; q >= 0; q--)
                            stList_append(lists[i], stIntTuple_construct3(tri[3 * q], tri[3 * q + 1] + origin[k].x1,
                                                                          tri[3 * q + 2] + origin[k].y1));
                    }
                    free(tri);
                }
            }
            free(np);
        } else if (dna) {
            /* the device's per-model sums go through the Hmm's own add functions (cell_updateExpectations
             * :407-424 calls them per transition) */
            Hmm *hmm = hmmOut;
            double e[CPECAN_EXPECTATION5_LEN];
            for (int32_t k = 0; k < nModels; k++) {
                CHECK(cpecan_hip_batch_fetch_expectations(batch, ids[k], e));
                for (int64_t f = 0; f < 5; f++)
                    for (int64_t t = 0; t < 5; t++)
                        if (e[f * 5 + t] != 0.0) hmm->addToTransitionExpectationFcn(hmm, f, t, e[f * 5 + t]);
                for (int64_t st = 0; st < 5; st++)
                    for (int64_t x = 0; x < 4; x++)
                        for (int64_t y = 0; y < 4; y++)
                            if (e[25 + st * 16 + x * 4 + y] != 0.0)
                                hmm->addToEmissionExpectationFcn(hmm, st, x, y, e[25 + st * 16 + x * 4 + y]);
                hmm->likelihood += e[CPECAN_EXPECTATION5_LEN - 1];
            }
        } else if (hdp) {
            HdpHmmExpectations *hmm = hmmOut;
            double e[CPECAN_EXPECTATIONH_LEN];
            for (int32_t k = 0; k < nModels; k++) {
                CHECK(cpecan_hip_batch_fetch_expectations(batch, ids[k], e));
                for (int q = 0; q < 9; q++) hmm->transitions[q] += e[q];
                hmm->likelihood += e[9];
            }
            /* assignments, sub-alignment after sub-alignment as the reference walks them: (k-mer, event mean) */
            int64_t *np = malloc(sizeof(int64_t) * (size_t) nItems);
            CHECK(cpecan_hip_batch_counts(batch, np, NULL, NULL));
            for (int64_t i = 0; i < n; i++)
                for (int64_t k = firstItem[i]; k < firstItem[i + 1]; k++) {
                    int64_t *tri = malloc(sizeof(int64_t) * 3 * (size_t) (np[k] + 1));
                    CHECK(cpecan_hip_batch_fetch_pairs(batch, k, tri, NULL, np[k] + 1));
                    for (int64_t q = 0; q < np[k]; q++) {
                        const int64_t x = tri[3 * q + 1] + origin[k].x1, y = tri[3 * q + 2] + origin[k].y1;
                        if (hmm->numberOfAssignments == hmm->capacity || !hmm->assignmentXY) {
                            hmm->capacity = hmm->capacity ? 2 * hmm->capacity : 1024;
                            hmm->eventAssignments = realloc(hmm->eventAssignments, sizeof(double) * (size_t) hmm->capacity);
                            hmm->kmerAssignments = realloc(hmm->kmerAssignments, (size_t) hmm->capacity * (KMER_LENGTH + 1));
                            hmm->assignmentXY = realloc(hmm->assignmentXY, sizeof(int64_t) * 3 * (size_t) hmm->capacity);
                        }
                        hmm->assignmentXY[3 * hmm->numberOfAssignments] = x >= 0 ? x : 0;
                        hmm->assignmentXY[3 * hmm->numberOfAssignments + 1] = y;
                        hmm->assignmentXY[3 * hmm->numberOfAssignments + 2] = i;
                        char *dst = hmm->kmerAssignments + hmm->numberOfAssignments * (KMER_LENGTH + 1);
                        memcpy(dst, (const char *) sXs[i]->elements + (x >= 0 ? x : 0), KMER_LENGTH);
                        dst[KMER_LENGTH] = 0;
                        hmm->eventAssignments[hmm->numberOfAssignments++] = ((const double *) sYs[i]->elements)[3 * y];
                    }
                    free(tri);
                }
            free(np);
        } else if (van) {
            VanillaHmmExpectations *hmm = hmmOut;
            double e[CPECAN_EXPECTATIONV_LEN];
            for (int32_t k = 0; k < nModels; k++) {
                CHECK(cpecan_hip_batch_fetch_expectations(batch, ids[k], e));
                for (int q = 0; q < 60; q++) hmm->kmerSkipBins[q] += e[q];
                hmm->likelihood += e[60];
            }
        } else {
            ContinuousPairHmmExpectations *hmm = hmmOut;
            double *e = malloc(sizeof(double) * CPECAN_EXPECTATION_LEN);
            for (int32_t k = 0; k < nModels; k++) {
                CHECK(cpecan_hip_batch_fetch_expectations(batch, ids[k], e));
                for (int q = 0; q < 9; q++) hmm->transitions[q] += e[q];
                for (int q = 0; q < NUM_OF_KMERS; q++) hmm->individualKmerGapProbs[q] += e[9 + q];
                hmm->likelihood += e[9 + NUM_OF_KMERS];
            }
            free(e);
        }
        CHECK(cpecan_hip_batch_destroy(batch));
    } else if (mode == 0) {
        for (int64_t i = 0; i < n; i++) lists[i] = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    }
    free(chars); free(events); free(anchors); free(items); free(origin); free(firstItem);
    free(models); free(modelOf); free(ids); free(models5); free(owner5); free(ychars); free(modelsV); free(modelsH); free(models4);
}

stList *getAlignedPairsUsingAnchors(StateMachine *sM, Sequence *SsX, Sequence *SsY, stList *anchorPairs,
                                    PairwiseAlignmentParameters *p, DiagonalPosteriorProbFn fn,
                                    bool raggedL, bool raggedR) {
    if (fn != diagonalCalculationPosteriorMatchProbs)
        die("cpecan: the GPU path implements diagonalCalculationPosteriorMatchProbs only");
    stList *out = NULL;
    run_reads(1, &sM, &SsX, &SsY, &anchorPairs, p, raggedL, raggedR, 0, 0, &out, NULL);
    return out;
}

void getPosteriorProbsWithBanding(StateMachine *sM, stList *anchorPairs, Sequence *sX, Sequence *sY,
                                  PairwiseAlignmentParameters *p, bool raggedL, bool raggedR,
                                  DiagonalPosteriorProbFn fn, void *extraArgs) {
    if (fn == diagonalCalculation_Expectations) { /* extraArgs is the Hmm (:850) */
        if (sM->type != fiveState && sM->type != fiveStateAsymmetric)
            die("cpecan: expectations through this entry point take a 5-state machine and its Hmm");
        run_reads(1, &sM, &sX, &sY, &anchorPairs, p, raggedL, raggedR, 1, 2, NULL, extraArgs);
        return;
    }
    if (fn != diagonalCalculationPosteriorMatchProbs)
        die("cpecan: the GPU path implements diagonalCalculationPosteriorMatchProbs and "
            "diagonalCalculation_Expectations only");
    stList *dest = ((void **) extraArgs)[0], *out = NULL;
    run_reads(1, &sM, &sX, &sY, &anchorPairs, p, raggedL, raggedR, 0, 2, &out, NULL);
    for (int64_t i = 0; i < stList_length(out); i++) {
        stIntTuple *t = stList_get(out, i);
        stList_append(dest, stIntTuple_construct3(stIntTuple_get(t, 0), stIntTuple_get(t, 1), stIntTuple_get(t, 2)));
    }
    stList_destruct(out);
}

stList **getAlignedPairsUsingAnchorsBatch(int64_t n, StateMachine **sMs, Sequence **sXs, Sequence **sYs,
                                          stList **anchors, PairwiseAlignmentParameters *p, bool raggedL,
                                          bool raggedR) {
    stList **out = calloc((size_t) (n > 0 ? n : 1), sizeof(stList *));
    if (n > 0) run_reads(n, sMs, sXs, sYs, anchors, p, raggedL, raggedR, 0, 0, out, NULL);
    return out;
}

stList *getAlignedPairsWithoutBanding(StateMachine *sM, void *cX, void *cY, int64_t lX, int64_t lY,
                                      PairwiseAlignmentParameters *p, void *(*getXFcn)(void *, int64_t),
                                      void *(*getYFcn)(void *, int64_t), DiagonalPosteriorProbFn fn,
                                      bool raggedL, bool raggedR) {
    if (fn != diagonalCalculationPosteriorMatchProbs)
        die("cpecan: the GPU path implements diagonalCalculationPosteriorMatchProbs only");
    Sequence *sX = sequence_construct(lX, cX, getXFcn), *sY = sequence_construct(lY, cY, getYFcn);
    stList *out = NULL, *none = NULL;
    run_reads(1, &sM, &sX, &sY, &none, p, raggedL, raggedR, 0, 1, &out, NULL);
    sequence_sequenceDestroy(sX);
    sequence_sequenceDestroy(sY);
    return out;
}

void cpecan_getSignalExpectationsUsingAnchors(StateMachine *sM, ContinuousPairHmmExpectations *hmm, Sequence *SsX,
                                       Sequence *SsY, stList *anchorPairs, PairwiseAlignmentParameters *p,
                                       bool raggedL, bool raggedR) {
    run_reads(1, &sM, &SsX, &SsY, &anchorPairs, p, raggedL, raggedR, 1, 0, NULL, hmm);
}

void cpecan_pairHmmExpectations_normalize(ContinuousPairHmmExpectations *hmm) {
    for (int from = 0; from < 3; from++) { /* hmmDiscrete_normalize2, impl/discreteHmm.c:125-136 */
        double total = 0.0;
        for (int to = 0; to < 3; to++) total += hmm->transitions[from * 3 + to];
        for (int to = 0; to < 3; to++) hmm->transitions[from * 3 + to] = hmm->transitions[from * 3 + to] / total;
    }
    double total = 0.0;
    for (int i = 0; i < NUM_OF_KMERS; i++) total += hmm->individualKmerGapProbs[i];
    for (int i = 0; i < NUM_OF_KMERS; i++) hmm->individualKmerGapProbs[i] = hmm->individualKmerGapProbs[i] / total;
}

void cpecan_pairHmmExpectations_load(StateMachine *sM, ContinuousPairHmmExpectations *hmm) {
    StateMachine3 *s = (StateMachine3 *) sM;
    const double *t = hmm->transitions;
    s->TRANSITION_MATCH_CONTINUE = log(t[match * 3 + match]);
    s->TRANSITION_GAP_OPEN_X = log(t[match * 3 + shortGapX]);
    s->TRANSITION_GAP_OPEN_Y = log(t[match * 3 + shortGapY]);
    s->TRANSITION_MATCH_FROM_GAP_X = log(t[shortGapX * 3 + match]);
    s->TRANSITION_GAP_EXTEND_X = log(1 - t[shortGapX * 3 + match]);
    s->TRANSITION_GAP_SWITCH_TO_Y = -INFINITY;
    s->TRANSITION_MATCH_FROM_GAP_Y = log(t[shortGapY * 3 + match]);
    s->TRANSITION_GAP_EXTEND_Y = log(t[shortGapY * 3 + shortGapY]);
    s->TRANSITION_GAP_SWITCH_TO_X = log(t[shortGapY * 3 + shortGapX]);
    for (int64_t i = 0; i < NUM_OF_KMERS; i++) sM->EMISSION_GAP_X_PROBS[i] = log(hmm->individualKmerGapProbs[i]);
}

/* ---- .hmm files of the strawMan expectations (impl/continuousHmm.c:234-370) --------------------------- */
void cpecan_pairHmmExpectations_write(ContinuousPairHmmExpectations *hmm, FILE *fh) {
    fprintf(fh, "%i\t%lld\t%lld\t\n", (int) threeState, 3ll, (long long) NUM_OF_KMERS);
    for (int i = 0; i < 9; i++)
        if (isnan(hmm->transitions[i])) { /* hmmContinuous_checkTransitions :48-58: nothing more is written */
            fprintf(stdout, "GOT NaN TRANS\n");
            return;
        }
    for (int i = 0; i < 9; i++) fprintf(fh, "%f\t", hmm->transitions[i]);
    fprintf(fh, "%f\n", hmm->likelihood);
    for (int i = 0; i < NUM_OF_KMERS; i++) fprintf(fh, "%f\t", hmm->individualKmerGapProbs[i]);
    fprintf(fh, "\n");
}
ContinuousPairHmmExpectations *cpecan_pairHmmExpectations_read(const char *fileName) {
    FILE *f = fopen(fileName, "r");
    if (!f) die("cpecan: cannot open %s", fileName);
    double hdr[3], line[10];
    if (line_doubles(f, hdr, 3) != 3) die("Failed to parse the header line of %s", fileName);
    if ((int) hdr[0] != (int) threeState || (int64_t) hdr[1] != 3 || (int64_t) hdr[2] != NUM_OF_KMERS)
        die("cpecan: %s is not a 3-state k-mer HMM (type %d, %lld states, %lld symbols)", fileName, (int) hdr[0],
            (long long) hdr[1], (long long) hdr[2]);
    ContinuousPairHmmExpectations *hmm = calloc(1, sizeof *hmm);
    const int64_t nT = line_doubles(f, line, 10);
    if (nT != 10)
        die("Incorrect number of transitions in the input HMM file %s, got %lld instead of %lld", fileName,
            (long long) nT, 10ll);
    memcpy(hmm->transitions, line, sizeof(double) * 9);
    hmm->likelihood = line[9];
    const int64_t nE = line_doubles(f, hmm->individualKmerGapProbs, NUM_OF_KMERS);
    if (nE != NUM_OF_KMERS)
        die("Incorrect number of emissions in the input HMM file %s, got %lld instead of %lld", fileName,
            (long long) nE, (long long) NUM_OF_KMERS);
    fclose(f);
    return hmm;
}

/* ---- HmmDiscrete (impl/discreteHmm.c) ----------------------------------------------------------------- */
Hmm *hmmDiscrete_constructEmpty(double pseudocount, int64_t stateNumber, int64_t symbolSetSize,
                                StateMachineType type,
                                void (*addToTransitionExpFcn)(Hmm *, int64_t, int64_t, double),
                                void (*setTransitionFcn)(Hmm *, int64_t, int64_t, double),
                                double (*getTransitionsExpFcn)(Hmm *, int64_t, int64_t),
                                void (*addEmissionsExpFcn)(Hmm *, int64_t, int64_t, int64_t, double),
                                void (*setEmissionExpFcn)(Hmm *, int64_t, int64_t, int64_t, double),
                                double (*getEmissionExpFcn)(Hmm *, int64_t, int64_t, int64_t),
                                int64_t (*getElementIndexFcn)(void *)) {
    HmmDiscrete *h = calloc(1, sizeof *h);
    h->baseHmm.stateNumber = stateNumber;
    h->baseHmm.symbolSetSize = symbolSetSize;
    h->baseHmm.matrixSize = symbolSetSize * symbolSetSize;
    h->baseHmm.type = type;
    const int64_t nT = stateNumber * stateNumber, nE = stateNumber * h->baseHmm.matrixSize;
    h->transitions = malloc(sizeof(double) * (size_t) nT);
    h->emissions = malloc(sizeof(double) * (size_t) nE);
    for (int64_t i = 0; i < nT; i++) h->transitions[i] = pseudocount;
    for (int64_t i = 0; i < nE; i++) h->emissions[i] = pseudocount;
    h->baseHmm.likelihood = 0.0;
    h->baseHmm.addToTransitionExpectationFcn = addToTransitionExpFcn;
    h->baseHmm.setTransitionFcn = setTransitionFcn;
    h->baseHmm.getTransitionsExpFcn = getTransitionsExpFcn;
    h->baseHmm.addToEmissionExpectationFcn = addEmissionsExpFcn;
    h->baseHmm.setEmissionExpectationFcn = setEmissionExpFcn;
    h->baseHmm.getEmissionExpFcn = getEmissionExpFcn;
    h->baseHmm.getElementIndexFcn = getElementIndexFcn;
    return (Hmm *) h;
}
#define HD(hmm) ((HmmDiscrete *) (hmm))
void hmmDiscrete_addToTransitionExpectation(Hmm *hmm, int64_t from, int64_t to, double p) {
    HD(hmm)->transitions[from * hmm->stateNumber + to] += p;
}
void hmmDiscrete_setTransitionExpectation(Hmm *hmm, int64_t from, int64_t to, double p) {
    HD(hmm)->transitions[from * hmm->stateNumber + to] = p;
}
double hmmDiscrete_getTransitionExpectation(Hmm *hmm, int64_t from, int64_t to) {
    return HD(hmm)->transitions[from * hmm->stateNumber + to];
}
void hmmDiscrete_addToEmissionExpectation(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p) {
    HD(hmm)->emissions[state * hmm->matrixSize + x * hmm->symbolSetSize + y] += p;
}
void hmmDiscrete_setEmissionExpectation(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p) {
    HD(hmm)->emissions[state * hmm->matrixSize + x * hmm->symbolSetSize + y] = p;
}
double hmmDiscrete_getEmissionExpectation(Hmm *hmm, int64_t state, int64_t x, int64_t y) {
    return HD(hmm)->emissions[state * hmm->matrixSize + x * hmm->symbolSetSize + y];
}
void hmmDiscrete_randomizeTransitions(Hmm *hmm) { /* st_random(): uniform on [0, 1) */
    for (int64_t from = 0; from < hmm->stateNumber; from++)
        for (int64_t to = 0; to < hmm->stateNumber; to++) hmm->setTransitionFcn(hmm, from, to, drand48());
}
void hmmDiscrete_randomizeEmissions(Hmm *hmm) {
    if (hmm->symbolSetSize <= 0) die("hmmDiscrete_randomizeEmissions: got NULL for symbolSetSize");
    for (int64_t s = 0; s < hmm->stateNumber; s++)
        for (int64_t x = 0; x < hmm->symbolSetSize; x++)
            for (int64_t y = 0; y < hmm->symbolSetSize; y++) hmm->setEmissionExpectationFcn(hmm, s, x, y, drand48());
}
void hmmDiscrete_randomize(Hmm *hmm) {
    hmmDiscrete_randomizeTransitions(hmm);
    hmmDiscrete_randomizeEmissions(hmm);
    hmmDiscrete_normalize2(hmm, true);
}
void hmmDiscrete_normalize(Hmm *hmm) { hmmDiscrete_normalize2(hmm, true); }
void hmmDiscrete_normalize2(Hmm *hmm, bool normalizeEmissions) {
    for (int64_t from = 0; from < hmm->stateNumber; from++) {
        double total = 0.0;
        for (int64_t to = 0; to < hmm->stateNumber; to++) total += hmm->getTransitionsExpFcn(hmm, from, to);
        for (int64_t to = 0; to < hmm->stateNumber; to++)
            hmm->setTransitionFcn(hmm, from, to, hmm->getTransitionsExpFcn(hmm, from, to) / total);
    }
    if (!normalizeEmissions) return;
    for (int64_t s = 0; s < hmm->stateNumber; s++) {
        double total = 0.0;
        for (int64_t x = 0; x < hmm->symbolSetSize; x++)
            for (int64_t y = 0; y < hmm->symbolSetSize; y++) total += hmm->getEmissionExpFcn(hmm, s, x, y);
        for (int64_t x = 0; x < hmm->symbolSetSize; x++)
            for (int64_t y = 0; y < hmm->symbolSetSize; y++)
                hmm->setEmissionExpectationFcn(hmm, s, x, y, hmm->getEmissionExpFcn(hmm, s, x, y) / total);
    }
}
void hmmDiscrete_write(Hmm *hmm, FILE *fh) {
    HmmDiscrete *h = HD(hmm);
    fprintf(fh, "%i\t%lld\t%lld\t\n", (int) hmm->type, (long long) hmm->stateNumber, (long long) hmm->symbolSetSize);
    for (int64_t i = 0; i < hmm->stateNumber * hmm->stateNumber; i++) fprintf(fh, "%f\t", h->transitions[i]);
    fprintf(fh, "%f\n", hmm->likelihood);
    for (int64_t i = 0; i < hmm->stateNumber * hmm->matrixSize; i++) fprintf(fh, "%f\t", h->emissions[i]);
    fprintf(fh, "\n");
}
Hmm *hmmDiscrete_loadFromFile(const char *fileName) {
    FILE *f = fopen(fileName, "r");
    if (!f) die("cpecan: cannot open %s", fileName);
    double hdr[3];
    if (line_doubles(f, hdr, 3) < 3) die("Got an empty line in the input state machine file %s", fileName);
    Hmm *hmm = hmmDiscrete_constructEmpty(0.0, (int64_t) hdr[1], (int64_t) hdr[2], (StateMachineType) (int) hdr[0],
                                          hmmDiscrete_addToTransitionExpectation,
                                          hmmDiscrete_setTransitionExpectation,
                                          hmmDiscrete_getTransitionExpectation,
                                          hmmDiscrete_addToEmissionExpectation,
                                          hmmDiscrete_setEmissionExpectation,
                                          hmmDiscrete_getEmissionExpectation, emissions_discrete_getBaseIndex);
    HmmDiscrete *h = HD(hmm);
    const int64_t nT = hmm->stateNumber * hmm->stateNumber, nE = hmm->stateNumber * hmm->matrixSize;
    double *line = malloc(sizeof(double) * (size_t) (nT + 1));
    const int64_t gotT = line_doubles(f, line, nT + 1);
    if (gotT != nT + 1) /* the likelihood ends the transition line */
        die("Got the wrong number of transitions in the input state machine file %s, got %lld instead of %lld",
            fileName, (long long) gotT, (long long) (nT + 1));
    memcpy(h->transitions, line, sizeof(double) * (size_t) nT);
    hmm->likelihood = line[nT];
    free(line);
    const int64_t gotE = line_doubles(f, h->emissions, nE);
    if (gotE != nE)
        die("Got the wrong number of emissions in the input state machine file %s, got %lld instead of %lld",
            fileName, (long long) gotE, (long long) nE);
    fclose(f);
    return hmm;
}
void hmmDiscrete_destruct(Hmm *hmm) {
    free(HD(hmm)->transitions);
    free(HD(hmm)->emissions);
    free(hmm);
}
int64_t emissions_discrete_getBaseIndex(void *base) {
    switch (*(char *) base) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return NUM_OF_KMERS + 1;
    }
}
StateMachineFunctions *stateMachineFunctions_construct(double (*gapXProbFcn)(const double *, void *),
                                                       double (*gapYProbFcn)(const double *, void *),
                                                       double (*matchProbFcn)(const double *, void *, void *)) {
    StateMachineFunctions *f = malloc(sizeof *f);
    f->gapXProbFcn = gapXProbFcn;
    f->gapYProbFcn = gapYProbFcn;
    f->matchProbFcn = matchProbFcn;
    return f;
}

/* ---- the M-step for the 5-state machine (impl/stateMachine.c:679-732, 1051-1154, 1698-1723) ------------ */
static void swap_doubles(double *a, double *b) { double t = *a; *a = *b; *b = t; }
static void em_load_gap_probs(double *gap, Hmm *hmm, const int64_t *xStates, int nX, const int64_t *yStates, int nY) {
    const int64_t n = hmm->symbolSetSize;
    for (int64_t i = 0; i < n; i++) gap[i] = 0.0;
    for (int k = 0; k < nX; k++) /* collapse [x][y] onto x */
        for (int64_t x = 0; x < n; x++)
            for (int64_t y = 0; y < n; y++) gap[x] += hmm->getEmissionExpFcn(hmm, xStates[k], x, y);
    for (int k = 0; k < nY; k++) /* ... onto y */
        for (int64_t x = 0; x < n; x++)
            for (int64_t y = 0; y < n; y++) gap[y] += hmm->getEmissionExpFcn(hmm, yStates[k], x, y);
    double total = 0.0;
    for (int64_t i = 0; i < n; i++) total += gap[i];
    for (int64_t i = 0; i < n; i++) gap[i] = log(gap[i] / total);
}
#define TR(f, t) hmm->getTransitionsExpFcn(hmm, f, t)
static void swap_short_long_x(StateMachine5 *s) {
    if (s->TRANSITION_GAP_SHORT_EXTEND_X > s->TRANSITION_GAP_LONG_EXTEND_X) {
        /* a "long" state that extends less than the "short" one (can happen during training): trade places */
        swap_doubles(&s->TRANSITION_GAP_SHORT_EXTEND_X, &s->TRANSITION_GAP_LONG_EXTEND_X);
        swap_doubles(&s->TRANSITION_MATCH_FROM_SHORT_GAP_X, &s->TRANSITION_MATCH_FROM_LONG_GAP_X);
        swap_doubles(&s->TRANSITION_GAP_SHORT_OPEN_X, &s->TRANSITION_GAP_LONG_OPEN_X);
        swap_doubles(&s->TRANSITION_GAP_SHORT_SWITCH_TO_X, &s->TRANSITION_GAP_LONG_SWITCH_TO_X);
    }
}
static void sm5_load_asymmetric(StateMachine5 *s, Hmm *hmm) {
    const int64_t n = hmm->symbolSetSize;
    s->TRANSITION_MATCH_CONTINUE = log(TR(match, match));
    s->TRANSITION_MATCH_FROM_SHORT_GAP_X = log(TR(shortGapX, match));
    s->TRANSITION_MATCH_FROM_LONG_GAP_X = log(TR(longGapX, match));
    s->TRANSITION_GAP_SHORT_OPEN_X = log(TR(match, shortGapX));
    s->TRANSITION_GAP_SHORT_EXTEND_X = log(TR(shortGapX, shortGapX));
    s->TRANSITION_GAP_SHORT_SWITCH_TO_X = log(TR(shortGapY, shortGapX));
    s->TRANSITION_GAP_LONG_OPEN_X = log(TR(match, longGapX));
    s->TRANSITION_GAP_LONG_EXTEND_X = log(TR(longGapX, longGapX));
    s->TRANSITION_GAP_LONG_SWITCH_TO_X = log(TR(longGapY, longGapX));
    swap_short_long_x(s);
    s->TRANSITION_MATCH_FROM_SHORT_GAP_Y = log(TR(shortGapY, match));
    s->TRANSITION_MATCH_FROM_LONG_GAP_Y = log(TR(longGapY, match));
    s->TRANSITION_GAP_SHORT_OPEN_Y = log(TR(match, shortGapY));
    s->TRANSITION_GAP_SHORT_EXTEND_Y = log(TR(shortGapY, shortGapY));
    s->TRANSITION_GAP_SHORT_SWITCH_TO_Y = log(TR(shortGapX, shortGapY));
    s->TRANSITION_GAP_LONG_OPEN_Y = log(TR(match, longGapY));
    s->TRANSITION_GAP_LONG_EXTEND_Y = log(TR(longGapY, longGapY));
    s->TRANSITION_GAP_LONG_SWITCH_TO_Y = log(TR(longGapX, longGapY));
    if (s->TRANSITION_GAP_SHORT_EXTEND_Y > s->TRANSITION_GAP_LONG_EXTEND_Y) {
        swap_doubles(&s->TRANSITION_GAP_SHORT_EXTEND_Y, &s->TRANSITION_GAP_LONG_EXTEND_Y);
        swap_doubles(&s->TRANSITION_MATCH_FROM_SHORT_GAP_Y, &s->TRANSITION_MATCH_FROM_LONG_GAP_Y);
        swap_doubles(&s->TRANSITION_GAP_SHORT_OPEN_Y, &s->TRANSITION_GAP_LONG_OPEN_Y);
        swap_doubles(&s->TRANSITION_GAP_SHORT_SWITCH_TO_Y, &s->TRANSITION_GAP_LONG_SWITCH_TO_Y);
    }
    for (int64_t x = 0; x < n; x++)
        for (int64_t y = 0; y < n; y++)
            s->model.EMISSION_MATCH_PROBS[x * n + y] = log(hmm->getEmissionExpFcn(hmm, match, x, y));
    const int64_t xs[2] = { shortGapX, longGapX }, ys[2] = { shortGapY, longGapY };
    em_load_gap_probs(s->model.EMISSION_GAP_X_PROBS, hmm, xs, 2, NULL, 0);
    em_load_gap_probs(s->model.EMISSION_GAP_Y_PROBS, hmm, NULL, 0, ys, 2);
}
static void sm5_load_symmetric(StateMachine5 *s, Hmm *hmm) {
    const int64_t n = hmm->symbolSetSize;
    s->TRANSITION_MATCH_CONTINUE = log(TR(match, match));
    s->TRANSITION_MATCH_FROM_SHORT_GAP_X = log((TR(shortGapX, match) + TR(shortGapY, match)) / 2);
    s->TRANSITION_MATCH_FROM_LONG_GAP_X = log((TR(longGapX, match) + TR(longGapY, match)) / 2);
    s->TRANSITION_GAP_SHORT_OPEN_X = log((TR(match, shortGapX) + TR(match, shortGapY)) / 2);
    s->TRANSITION_GAP_SHORT_EXTEND_X = log((TR(shortGapX, shortGapX) + TR(shortGapY, shortGapY)) / 2);
    s->TRANSITION_GAP_SHORT_SWITCH_TO_X = log((TR(shortGapX, shortGapY) + TR(shortGapY, shortGapX)) / 2);
    s->TRANSITION_GAP_LONG_OPEN_X = log((TR(match, longGapX) + TR(match, longGapY)) / 2);
    s->TRANSITION_GAP_LONG_EXTEND_X = log((TR(longGapX, longGapX) + TR(longGapY, longGapY)) / 2);
    s->TRANSITION_GAP_LONG_SWITCH_TO_X = log((TR(longGapX, longGapY) + TR(longGapY, longGapX)) / 2);
    swap_short_long_x(s);
    s->TRANSITION_MATCH_FROM_SHORT_GAP_Y = s->TRANSITION_MATCH_FROM_SHORT_GAP_X;
    s->TRANSITION_MATCH_FROM_LONG_GAP_Y = s->TRANSITION_MATCH_FROM_LONG_GAP_X;
    s->TRANSITION_GAP_SHORT_OPEN_Y = s->TRANSITION_GAP_SHORT_OPEN_X;
    s->TRANSITION_GAP_SHORT_EXTEND_Y = s->TRANSITION_GAP_SHORT_EXTEND_X;
    s->TRANSITION_GAP_SHORT_SWITCH_TO_Y = s->TRANSITION_GAP_SHORT_SWITCH_TO_X;
    s->TRANSITION_GAP_LONG_OPEN_Y = s->TRANSITION_GAP_LONG_OPEN_X;
    s->TRANSITION_GAP_LONG_EXTEND_Y = s->TRANSITION_GAP_LONG_EXTEND_X;
    s->TRANSITION_GAP_LONG_SWITCH_TO_Y = s->TRANSITION_GAP_LONG_SWITCH_TO_X;
    for (int64_t x = 0; x < n; x++) {
        s->model.EMISSION_MATCH_PROBS[x * n + x] = log(hmm->getEmissionExpFcn(hmm, match, x, x));
        for (int64_t y = x + 1; y < n; y++) {
            const double d = log((hmm->getEmissionExpFcn(hmm, match, x, y) + hmm->getEmissionExpFcn(hmm, match, y, x)) / 2.0);
            s->model.EMISSION_MATCH_PROBS[x * n + y] = d;
            s->model.EMISSION_MATCH_PROBS[y * n + x] = d;
        }
    }
    const int64_t xs[2] = { shortGapX, longGapX }, ys[2] = { shortGapY, longGapY };
    em_load_gap_probs(s->model.EMISSION_GAP_X_PROBS, hmm, xs, 2, ys, 2);
    em_load_gap_probs(s->model.EMISSION_GAP_Y_PROBS, hmm, xs, 2, ys, 2);
}
#undef TR
StateMachine *getStateMachine5(Hmm *hmmD, StateMachineFunctions *sMfs) {
    if (hmmD->type != fiveState && hmmD->type != fiveStateAsymmetric) die("Wrong hmm type");
    /* the reference constructs with type fiveState in both cases (:1700,:1710) and zeroed emissions */
    StateMachine5 *s = (StateMachine5 *) stateMachine5_construct(fiveState, hmmD->symbolSetSize, NULL,
                                                                 sMfs->gapXProbFcn, sMfs->gapYProbFcn,
                                                                 sMfs->matchProbFcn, cell_updateExpectations);
    if (hmmD->type == fiveState) sm5_load_symmetric(s, hmmD);
    else sm5_load_asymmetric(s, hmmD);
    return (StateMachine *) s;
}

/* The E-step of n reads as ONE batch on the GPU, added to the Hmm through its own add functions -- what
 * cell_updateExpectations (:407-424) and cell_signal_updateTransAndKmerSkipExpectations(2) (:426-476) do one exp() at
 * a time.  Every read's state machine must be of the Hmm's type. */
static void e_step_into_hmm(int64_t n, StateMachine **sMs, Hmm *hmm, Sequence **sXs, Sequence **sYs, stList **anchors,
                            PairwiseAlignmentParameters *p, bool raggedL, bool raggedR) {
    if (n <= 0) return;
    for (int64_t i = 0; i < n; i++)
        if (hmm->type != sMs[i]->type && !(sMs[i]->type == fiveState && hmm->type == fiveStateAsymmetric))
            die("cpecan: getExpectationsUsingAnchors: the Hmm (type %d) does not belong to the state machine (type %d)",
                (int) hmm->type, (int) sMs[i]->type);
    switch (sMs[0]->type) {
    case fiveState:
    case fiveStateAsymmetric:
        if (hmm->stateNumber != 5 || hmm->symbolSetSize != SYMBOL_NUMBER_NO_N)
            die("cpecan: the 5-state E-step takes a 5-state, 4-symbol Hmm");
        run_reads(n, sMs, sXs, sYs, anchors, p, raggedL, raggedR, 1, 0, NULL, hmm);
        return;
    case threeState: {
        ContinuousPairHmmExpectations *e = calloc(1, sizeof *e);
        run_reads(n, sMs, sXs, sYs, anchors, p, raggedL, raggedR, 1, 0, NULL, e);
        hmm->likelihood += e->likelihood;
        for (int64_t from = 0; from < 3; from++)
            for (int64_t to = 0; to < 3; to++)
                hmm->addToTransitionExpectationFcn(hmm, from, to, e->transitions[from * 3 + to]);
        for (int64_t k = 0; k < NUM_OF_KMERS; k++)
            if (e->individualKmerGapProbs[k] != 0.0)
                hmm->addToEmissionExpectationFcn(hmm, 0, k, 0, e->individualKmerGapProbs[k]);
        free(e);
        return;
    }
    case vanilla: {
        VanillaHmmExpectations e;
        memset(&e, 0, sizeof e);
        run_reads(n, sMs, sXs, sYs, anchors, p, raggedL, raggedR, 1, 0, NULL, &e);
        hmm->likelihood += e.likelihood;
        for (int64_t bin = 0; bin < 60; bin++) hmm->addToTransitionExpectationFcn(hmm, bin, 0, e.kmerSkipBins[bin]);
        return;
    }
    case threeStateHdp: {
        HdpHmm *h = (HdpHmm *) hmm;
        HdpHmmExpectations *e = cpecan_hdpExpectations_construct(0.0, h->threshold);
        run_reads(n, sMs, sXs, sYs, anchors, p, raggedL, raggedR, 1, 0, NULL, e);
        hmm->likelihood += e->likelihood;
        for (int64_t from = 0; from < 3; from++)
            for (int64_t to = 0; to < 3; to++)
                hmm->addToTransitionExpectationFcn(hmm, from, to, e->transitions[from * 3 + to]);
        /* as the reference's cell_signal_updateTransAndKmerSkipExpectations2 does: pointers into SsX and SsY */
        for (int64_t i = 0; i < e->numberOfAssignments; i++) {
            const int64_t r = e->assignmentXY[3 * i + 2];
            h->addToAssignments(hmm, (char *) sXs[r]->elements + e->assignmentXY[3 * i],
                                (double *) sYs[r]->elements + NB_EVENT_PARAMS * e->assignmentXY[3 * i + 1]);
        }
        cpecan_hdpExpectations_destruct(e);
        return;
    }
    default:
        die("cpecan: getExpectationsUsingAnchors: no E-step for state machine type %d", (int) sMs[0]->type);
    }
}
void getExpectationsUsingAnchors(StateMachine *sM, Hmm *hmmExpectations, Sequence *SsX, Sequence *SsY,
                                 stList *anchorPairs, PairwiseAlignmentParameters *p,
                                 DiagonalPosteriorProbFn fn, bool raggedL, bool raggedR) {
    if (fn != diagonalCalculation_Expectations)
        die("cpecan: the GPU path implements diagonalCalculation_Expectations only");
    e_step_into_hmm(1, &sM, hmmExpectations, &SsX, &SsY, &anchorPairs, p, raggedL, raggedR);
}
void getExpectationsUsingAnchorsBatch(int64_t n, StateMachine **sMs, Hmm *hmmExpectations, Sequence **sXs,
                                      Sequence **sYs, stList **anchors, PairwiseAlignmentParameters *p,
                                      bool raggedL, bool raggedR) {
    e_step_into_hmm(n, sMs, hmmExpectations, sXs, sYs, anchors, p, raggedL, raggedR);
}

/* ---- the training loop (scripts/trainModels.py:244-330 for the signal machines, cPecanEm.py:107-209 for the
 * discrete one) as one native call: per iteration an empty Hmm with pseudocounts, the E-step of this rank's reads as
 * one GPU batch, the ranks' expectations summed (reduce), the normalisation, and the new parameters loaded into every
 * read's state machine; the running likelihood is logged per iteration. ------------------------------------------- */
static Hmm *empty_hmm_for(StateMachineType type, double pseudocount, double threshold) {
    if (type == fiveState || type == fiveStateAsymmetric)
        return hmmDiscrete_constructEmpty(pseudocount, 5, SYMBOL_NUMBER_NO_N, type, hmmDiscrete_addToTransitionExpectation,
                                          hmmDiscrete_setTransitionExpectation, hmmDiscrete_getTransitionExpectation,
                                          hmmDiscrete_addToEmissionExpectation, hmmDiscrete_setEmissionExpectation,
                                          hmmDiscrete_getEmissionExpectation, emissions_discrete_getBaseIndex);
    return hmmContinuous_getEmptyHmm(type, pseudocount, threshold);
}
static void destroy_hmm_of(Hmm *hmm) {
    if (hmm->type == fiveState || hmm->type == fiveStateAsymmetric) hmmDiscrete_destruct(hmm);
    else hmmContinuous_destruct(hmm, hmm->type);
}
/* the expectation values of an Hmm as one vector (for the all-reduce) and back: likelihood first, then the
 * "transition" table (the vanilla machine keeps its 60 skip bins there), then the emission table it has */
static int64_t hmm_vector(Hmm *hmm, double *v, bool store) {
    int64_t k = 0;
    if (v) { if (store) hmm->likelihood = v[k]; else v[k] = hmm->likelihood; }
    k++;
    if (hmm->type == vanilla) {
        for (int64_t bin = 0; bin < 60; bin++, k++)
            if (v) { if (store) hmm->setTransitionFcn(hmm, bin, 0, v[k]); else v[k] = hmm->getTransitionsExpFcn(hmm, bin, 0); }
        return k;
    }
    for (int64_t from = 0; from < hmm->stateNumber; from++)
        for (int64_t to = 0; to < hmm->stateNumber; to++, k++)
            if (v) { if (store) hmm->setTransitionFcn(hmm, from, to, v[k]); else v[k] = hmm->getTransitionsExpFcn(hmm, from, to); }
    if (hmm->type == threeState)
        for (int64_t i = 0; i < hmm->symbolSetSize; i++, k++)
            if (v) { if (store) hmm->setEmissionExpectationFcn(hmm, 0, i, 0, v[k]); else v[k] = hmm->getEmissionExpFcn(hmm, 0, i, 0); }
    if (hmm->type == fiveState || hmm->type == fiveStateAsymmetric)
        for (int64_t st = 0; st < hmm->stateNumber; st++)
            for (int64_t x = 0; x < hmm->symbolSetSize; x++)
                for (int64_t y = 0; y < hmm->symbolSetSize; y++, k++)
                    if (v) { if (store) hmm->setEmissionExpectationFcn(hmm, st, x, y, v[k]); else v[k] = hmm->getEmissionExpFcn(hmm, st, x, y); }
    return k;
}
Hmm *cpecan_trainModels(int64_t nReads, StateMachine **sMs, Sequence **sXs, Sequence **sYs, stList **anchorPairs,
                        PairwiseAlignmentParameters *p, bool raggedL, bool raggedR, StateMachineType hmmType,
                        int64_t iterations, double pseudocount, double hdpThreshold, cpecan_reduce_fn reduce,
                        void *reduceArg, double *runningLikelihoods) {
    if (nReads < 0 || iterations < 0 || (nReads > 0 && (!sMs || !sXs || !sYs || !anchorPairs)) || !p)
        die("cpecan_trainModels: bad argument");
    Hmm *hmm = NULL;
    for (int64_t it = 0; it < iterations || !hmm; it++) {
        if (hmm) destroy_hmm_of(hmm);
        hmm = empty_hmm_for(hmmType, pseudocount, hdpThreshold);
        if (iterations == 0) break;
        e_step_into_hmm(nReads, sMs, hmm, sXs, sYs, anchorPairs, p, raggedL, raggedR);
        if (reduce) {
            const int64_t len = hmm_vector(hmm, NULL, false);
            double *v = malloc(sizeof(double) * (size_t) len);
            hmm_vector(hmm, v, false);
            reduce(reduceArg, v, len);
            hmm_vector(hmm, v, true);
            free(v);
        }
        if (runningLikelihoods) runningLikelihoods[it] = hmm->likelihood;
        /* the M-step, and the new parameters into every (distinct) state machine of the reads */
        if (hmm->type == fiveState || hmm->type == fiveStateAsymmetric) hmmDiscrete_normalize2(hmm, true);
        else if (hmm->type == threeStateHdp) hmmDiscrete_normalize2(hmm, false);
        else hmmContinuous_normalize(hmm, hmm->type);
        for (int64_t i = 0; i < nReads; i++) {
            bool seen = false;
            for (int64_t k = 0; k < i && !seen; k++) seen = sMs[k] == sMs[i];
            if (seen) continue;
            if (hmm->type == fiveState) sm5_load_symmetric((StateMachine5 *) sMs[i], hmm);
            else if (hmm->type == fiveStateAsymmetric) sm5_load_asymmetric((StateMachine5 *) sMs[i], hmm);
            else if (hmm->type == threeState) continuousPairHmm_loadTransitionsAndKmerGapProbs(sMs[i], hmm);
            else if (hmm->type == vanilla) vanillaHmm_loadKmerSkipBinExpectations(sMs[i], hmm);
            else hdpHmm_loadTransitions(sMs[i], hmm);
        }
    }
    return hmm;
}

void getExpectations(StateMachine *sM, Hmm *hmmExpectations, void *sX, void *sY, int64_t lX, int64_t lY,
                     PairwiseAlignmentParameters *p, void *(*getFcn)(void *, int64_t),
                     stList *(*getAnchorPairFcn)(void *, void *, PairwiseAlignmentParameters *),
                     bool raggedL, bool raggedR) {
    stList *anchorPairs = getAnchorPairFcn(sX, sY, p);
    Sequence *SsX = sequence_construct2(lX, sX, getFcn, sequence_sliceNucleotideSequence2);
    Sequence *SsY = sequence_construct2(lY, sY, getFcn, sequence_sliceNucleotideSequence2);
    getExpectationsUsingAnchors(sM, hmmExpectations, SsX, SsY, anchorPairs, p, diagonalCalculation_Expectations,
                                raggedL, raggedR);
    sequence_sequenceDestroy(SsX);
    sequence_sequenceDestroy(SsY);
    stList_destruct(anchorPairs);
}

/* ---- anchor generation (impl/pairwiseAligner.c:1065-1281): host code around an EXTERNAL aligner.  As in the
 * reference, the anchors come from lastz run through a pipe (the reference calls "./cPecanLastz", built from its
 * externalTools/; CPECAN_LASTZ names another executable) and read back in exonerate CIGAR format; nothing of this
 * runs on the GPU.  Callers that bring a guide alignment (vanillaAlign) never come here. -------------------------- */
static char *temp_fasta(const char *name, const char *seq, bool upper) {
    char *path = malloc(64);
    strcpy(path, "/tmp/cpecan_lastz_XXXXXX");
    const int fd = mkstemp(path);
    if (fd < 0) die("cpecan: getBlastPairs: cannot create a temporary file");
    FILE *f = fdopen(fd, "w");
    fprintf(f, ">%s\n", name);
    for (const char *c = seq; *c; c++) fputc(upper ? toupper((unsigned char) *c) : *c, f);
    fputc('\n', f);
    fclose(f);
    return path;
}
/* one "cigar: <seq2> s e strand <seq1> s e strand score (op length)*" line into anchor pairs: the columns of its
 * match operations, trimmed by `trim` at both ends (convertPairwiseForwardStrandAlignmentToAnchorPairs :1036-1063) */
static void cigar_line_to_anchor_pairs(char *line, int64_t trim, stList *out) {
    char c2[256], c1[256], st2, st1;
    long long s2, e2, s1, e1;
    double score;
    int used = 0;
    if (sscanf(line, "cigar: %255s %lld %lld %c %255s %lld %lld %c %lf%n", c2, &s2, &e2, &st2, c1, &s1, &e1, &st1, &score,
               &used) != 9)
        return;
    if (strcmp(c1, "a") != 0 || strcmp(c2, "b") != 0 || st1 != '+' || st2 != '+')
        die("cpecan: getBlastPairs: unexpected alignment line from lastz: %s", line);
    int64_t j = s1, k = s2;
    char *q = line + used;
    for (;;) {
        char op;
        long long len;
        int n = 0;
        if (sscanf(q, " %c %lld%n", &op, &len, &n) != 2) break;
        q += n;
        if (op == 'M')
            for (int64_t l = trim; l < len - trim; l++) stList_append(out, stIntTuple_construct2(j + l, k + l));
        if (op != 'I') j += len; /* 'I': sequence 2 only (PAIRWISE_INDEL_Y) */
        if (op != 'D') k += len; /* 'D': sequence 1 only (PAIRWISE_INDEL_X) */
    }
    if (j != e1 || k != e2) die("cpecan: getBlastPairs: the operations of an alignment line do not add up: %s", line);
}
stList *getBlastPairs(const char *sX, const char *sY, int64_t trim, bool repeatMask) {
    stList *pairs = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    const size_t lX = strlen(sX), lY = strlen(sY);
    if (lX == 0 || lY == 0) return pairs;
    const char *lastz = getenv("CPECAN_LASTZ") ? getenv("CPECAN_LASTZ") : "./cPecanLastz";
    /* (the reference pipes a short Y through echo; two files either way here: same sequences, same options) */
    char *fa = temp_fasta("a", sX, !repeatMask), *fb = temp_fasta("b", sY, !repeatMask);
    const size_t room = strlen(lastz) + 256 + 2 * 64;
    char *command = malloc(room);
    snprintf(command, room, "%s --hspthresh=1800 --chain --strand=plus --gapped --format=cigar --gap=100,100 "
                            "--ambiguous=iupac,100,100 %s %s", lastz, fa, fb);
    FILE *fh = popen(command, "r");
    if (!fh) die("cpecan: getBlastPairs: problems with the lastz pipe (%s)", command);
    char *line = NULL;
    size_t cap = 0;
    while (getline(&line, &cap, fh) >= 0) cigar_line_to_anchor_pairs(line, trim, pairs);
    free(line);
    const int status = pclose(fh);
    remove(fa);
    remove(fb);
    if (status != 0) die("cpecan: getBlastPairs: '%s' ended with status %d", command, status);
    free(command);
    free(fa);
    free(fb);
    stList_sort(pairs, sortByXPlusYCoordinate); /* increasing coordinates */
    return pairs;
}
/* anchors inside a gap between top-level anchors that is still too large, without repeat masking (:1201-1226) */
static void blast_pairs_in_gap(const char *sX, const char *sY, int64_t pX, int64_t pY, int64_t x, int64_t y,
                               PairwiseAlignmentParameters *p, stList *combined) {
    const int64_t lX2 = x - pX, lY2 = y - pY;
    if (lX2 * lY2 <= p->repeatMaskMatrixBiggerThanThis) return;
    char *sX2 = malloc((size_t) lX2 + 1), *sY2 = malloc((size_t) lY2 + 1);
    memcpy(sX2, sX + pX, (size_t) lX2); sX2[lX2] = 0;
    memcpy(sY2, sY + pY, (size_t) lY2); sY2[lY2] = 0;
    stList *unfiltered = getBlastPairs(sX2, sY2, p->constraintDiagonalTrim, 0);
    stList_sort(unfiltered, stIntTuple_cmpFn);
    stList *bottom = filterToRemoveOverlap(unfiltered);
    stList_destruct(unfiltered);
    for (int64_t k = 0; k < stList_length(bottom); k++) {
        stIntTuple *t = stList_get(bottom, k);
        stList_append(combined, stIntTuple_construct2(stIntTuple_get(t, 0) + pX, stIntTuple_get(t, 1) + pY));
    }
    stList_destruct(bottom);
    free(sX2);
    free(sY2);
}
stList *getBlastPairsForPairwiseAlignmentParameters(void *sX, void *sY, PairwiseAlignmentParameters *p) {
    const char *cX = sX, *cY = sY;
    const int64_t lX = (int64_t) strlen(cX), lY = (int64_t) strlen(cY);
    if (lX * lY <= p->anchorMatrixBiggerThanThis) return stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    stList *unfiltered = getBlastPairs(cX, cY, p->constraintDiagonalTrim, 1);
    stList_sort(unfiltered, stIntTuple_cmpFn);
    stList *top = filterToRemoveOverlap(unfiltered);
    stList_destruct(unfiltered);
    stList *combined = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    int64_t pX = 0, pY = 0;
    for (int64_t i = 0; i < stList_length(top); i++) {
        stIntTuple *t = stList_get(top, i);
        const int64_t x = stIntTuple_get(t, 0), y = stIntTuple_get(t, 1);
        if (x < pX || y < pY || x >= lX || y >= lY) die("cpecan: getBlastPairs: anchors out of order or out of range");
        blast_pairs_in_gap(cX, cY, pX, pY, x, y, p, combined);
        stList_append(combined, stIntTuple_construct2(x, y));
        pX = x + 1;
        pY = y + 1;
    }
    blast_pairs_in_gap(cX, cY, pX, pY, lX, lY, p, combined);
    stList_destruct(top);
    return combined;
}

/* ---- re-weighting (impl/pairwiseAligner.c:1619-1667) -------------------------------------------------- */
int64_t *getIndelProbabilities(stList *alignedPairs, int64_t seqLength, bool xIfTrueElseY) {
    int64_t *gap = malloc(sizeof(int64_t) * (size_t) (seqLength > 0 ? seqLength : 1));
    for (int64_t i = 0; i < seqLength; i++) gap[i] = PAIR_ALIGNMENT_PROB_1;
    for (int64_t i = 0; i < stList_length(alignedPairs); i++) {
        stIntTuple *t = stList_get(alignedPairs, i);
        gap[stIntTuple_get(t, xIfTrueElseY ? 1 : 2)] -= stIntTuple_get(t, 0);
    }
    for (int64_t i = 0; i < seqLength; i++)
        if (gap[i] < 0) gap[i] = 0;
    return gap;
}
stList *reweightAlignedPairs(stList *alignedPairs, int64_t *indelProbsX, int64_t *indelProbsY, double gapGamma) {
    stList *out = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    for (int64_t i = 0; i < stList_length(alignedPairs); i++) {
        stIntTuple *t = stList_get(alignedPairs, i);
        const int64_t x = stIntTuple_get(t, 1), y = stIntTuple_get(t, 2);
        /* int64 - double, converted back on assignment, as the reference writes it */
        const int64_t w = stIntTuple_get(t, 0) - gapGamma * (indelProbsX[x] + indelProbsY[y]);
        stList_append(out, stIntTuple_construct3(w, x, y));
    }
    stList_destruct(alignedPairs);
    return out;
}
stList *reweightAlignedPairs2(stList *alignedPairs, int64_t seqLengthX, int64_t seqLengthY, double gapGamma) {
    if (gapGamma <= 0.0) return alignedPairs;
    int64_t *gx = getIndelProbabilities(alignedPairs, seqLengthX, 1);
    int64_t *gy = getIndelProbabilities(alignedPairs, seqLengthY, 0);
    alignedPairs = reweightAlignedPairs(alignedPairs, gx, gy, gapGamma);
    free(gx);
    free(gy);
    return alignedPairs;
}
void sequence_padSequence(Sequence *sequence) {
    const char *pad = "nnnnnnnnnnnnnnnnnnnnnnnnnnnnnn";
    const char *old = sequence->elements;
    char *padded = malloc(strlen(old) + strlen(pad) + 1);
    strcpy(padded, old);
    strcat(padded, pad);
    sequence->elements = padded;
}

/* ---- vanilla machine: E-step entry and the M-step of its skip bins (impl/continuousHmm.c:420-462) ------- */
void cpecan_getVanillaExpectationsUsingAnchors(StateMachine *sM, VanillaHmmExpectations *hmm, Sequence *SsX,
                                        Sequence *SsY, stList *anchorPairs, PairwiseAlignmentParameters *p,
                                        bool raggedL, bool raggedR) {
    if (sM->type != vanilla) die("cpecan: getVanillaExpectationsUsingAnchors takes a StateMachine3Vanilla");
    run_reads(1, &sM, &SsX, &SsY, &anchorPairs, p, raggedL, raggedR, 1, 0, NULL, hmm);
}
void cpecan_vanillaExpectations_normalize(VanillaHmmExpectations *hmm) {
    double total = 0.0; /* alpha and beta bins together, as the reference does */
    for (int i = 0; i < 60; i++) total += hmm->kmerSkipBins[i];
    for (int i = 0; i < 60; i++) hmm->kmerSkipBins[i] = hmm->kmerSkipBins[i] / total;
}
void cpecan_vanillaExpectations_load(StateMachine *sM, VanillaHmmExpectations *hmm) {
    if (sM->type != vanilla) die("you gave me the wrong type of HMM");
    for (int i = 0; i < 60; i++) sM->EMISSION_GAP_X_PROBS[i] = hmm->kmerSkipBins[i];
}

/* ---- HDP machine: E-step entry, transition M-step and the .expectations file (impl/continuousHmm.c:631-753) -- */
HdpHmmExpectations *cpecan_hdpExpectations_construct(double pseudocount, double threshold) {
    HdpHmmExpectations *h = calloc(1, sizeof *h);
    for (int i = 0; i < 9; i++) h->transitions[i] = pseudocount;
    h->threshold = threshold;
    return h;
}
void cpecan_hdpExpectations_destruct(HdpHmmExpectations *hmm) {
    free(hmm->eventAssignments);
    free(hmm->kmerAssignments);
    free(hmm->assignmentXY);
    free(hmm);
}
void cpecan_getHdpExpectationsUsingAnchors(StateMachine *sM, HdpHmmExpectations *hmm, Sequence *SsX, Sequence *SsY,
                                    stList *anchorPairs, PairwiseAlignmentParameters *p, bool raggedL, bool raggedR) {
    if (sM->type != threeStateHdp) die("cpecan: getHdpExpectationsUsingAnchors takes a StateMachine3_HDP");
    run_reads(1, &sM, &SsX, &SsY, &anchorPairs, p, raggedL, raggedR, 1, 0, NULL, hmm);
}
void cpecan_hdpExpectations_load(StateMachine *sM, HdpHmmExpectations *hmm) {
    StateMachine3_HDP *s = (StateMachine3_HDP *) sM;
    const double *t = hmm->transitions;
    s->TRANSITION_MATCH_CONTINUE = log(t[match * 3 + match]);
    s->TRANSITION_GAP_OPEN_X = log(t[match * 3 + shortGapX]);
    s->TRANSITION_GAP_OPEN_Y = log(t[match * 3 + shortGapY]);
    s->TRANSITION_MATCH_FROM_GAP_X = log(t[shortGapX * 3 + match]);
    s->TRANSITION_GAP_EXTEND_X = log(1 - t[shortGapX * 3 + match]); /* tied to the line above */
    s->TRANSITION_GAP_SWITCH_TO_Y = LOG_ZERO;                       /* no skip -> extra event */
    s->TRANSITION_MATCH_FROM_GAP_Y = log(t[shortGapY * 3 + match]);
    s->TRANSITION_GAP_EXTEND_Y = log(t[shortGapY * 3 + shortGapY]);
    s->TRANSITION_GAP_SWITCH_TO_X = log(t[shortGapY * 3 + shortGapX]);
}
void cpecan_hdpExpectations_write(HdpHmmExpectations *hmm, FILE *fh) {
    fprintf(fh, "%i\t%lld\t%lf\t%lld\t\n", (int) threeStateHdp, 3ll, hmm->threshold,
            (long long) hmm->numberOfAssignments);
    for (int i = 0; i < 9; i++)
        if (isnan(hmm->transitions[i])) {
            fprintf(stdout, "GOT NaN TRANS\n");
            return;
        }
    for (int i = 0; i < 9; i++) fprintf(fh, "%f\t", hmm->transitions[i]);
    fprintf(fh, "%f\n", hmm->likelihood);
    for (int64_t i = 0; i < hmm->numberOfAssignments; i++) fprintf(fh, "%lf\t", hmm->eventAssignments[i]);
    fprintf(fh, "\n");
    for (int64_t i = 0; i < hmm->numberOfAssignments; i++)
        fprintf(fh, "%.*s\t", KMER_LENGTH, hmm->kmerAssignments + i * (KMER_LENGTH + 1));
    fprintf(fh, "\n");
}

/* ---- the TSV of aligned pairs (vanillaAlign.c:26-96) -------------------------------------------------- */
static char complement_base(char c) { /* as stString_reverseComplementString does for nucleotides */
    switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
    default: return c;
    }
}
void writePosteriorProbs(char *posteriorProbsFile, char *readFile, double *matchModel, double scale, double shift,
                         double *events, char *target, bool forward, char *contig, int64_t eventSequenceOffset,
                         int64_t referenceSequenceOffset, stList *alignedPairs, Strand strand) {
    const char *strandLabel = strand == template ? "t" : "c";
    /* template read forward or complement read backward: the pairs index the reference as given */
    const bool sameSense = (strand == template && forward) || (strand == complement && !forward);
    FILE *fh = fopen(posteriorProbsFile, "a");
    if (!fh) die("cpecan: cannot open %s", posteriorProbsFile);
    const int64_t refLength = (int64_t) strlen(target);
    for (int64_t i = 0; i < stList_length(alignedPairs); i++) {
        stIntTuple *pair = stList_get(alignedPairs, i);
        const int64_t x = stIntTuple_get(pair, 1);
        const int64_t xAdj = sameSense ? x + referenceSequenceOffset
                                       : (refLength - KMER_LENGTH) - (x + (refLength - referenceSequenceOffset));
        const int64_t y = stIntTuple_get(pair, 2) + eventSequenceOffset;
        const double p = ((double) stIntTuple_get(pair, 0)) / PAIR_ALIGNMENT_PROB_1;
        const double mean = events[y * NB_EVENT_PARAMS], noise = events[y * NB_EVENT_PARAMS + 1],
                     duration = events[y * NB_EVENT_PARAMS + 2];
        char kmer[KMER_LENGTH + 1], refKmer[KMER_LENGTH + 1];
        memcpy(kmer, target + x, KMER_LENGTH);
        kmer[KMER_LENGTH] = 0;
        for (int k = 0; k < KMER_LENGTH; k++)
            refKmer[k] = sameSense ? kmer[k] : complement_base(kmer[KMER_LENGTH - 1 - k]);
        refKmer[KMER_LENGTH] = 0;
        const int64_t ki = emissions_discrete_getKmerIndex(kmer);
        const double levelMean = matchModel[1 + ki * MODEL_PARAMS], noiseMean = matchModel[1 + ki * MODEL_PARAMS + 2];
        fprintf(fh, "%s\t%lld\t%s\t%s\t%s\t%lld\t%f\t%f\t%f\t%s\t%f\t%f\t%f\t%f\t%f\n", contig,
                (long long) xAdj, refKmer, readFile, strandLabel, (long long) y, mean, noise, duration, kmer,
                levelMean, noiseMean, p, (mean - shift) / scale, (levelMean - shift) / scale);
    }
    fclose(fh);
}

/* ---- the split driver and getAlignedPairs, as the reference composes them (:1356-1422, :1486-1510) ---------- */
void getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps(
    StateMachine *sM, stList *anchorPairs, Sequence *SsX, Sequence *SsY, PairwiseAlignmentParameters *p, bool raggedL,
    bool raggedR, DiagonalPosteriorProbFn fn, void (*coordinateCorrectionFn)(int64_t, int64_t, void *),
    void *extraArgs) {
    stList *splitPoints = getSplitPoints(anchorPairs, SsX->length, SsY->length, p->splitMatrixBiggerThanThis,
                                         raggedL, raggedR);
    int64_t j = 0;
    const int64_t n = stList_length(splitPoints);
    for (int64_t i = 0; i < n; i++) {
        stIntTuple *r = stList_get(splitPoints, i);
        const int64_t x1 = stIntTuple_get(r, 0), y1 = stIntTuple_get(r, 1), x2 = stIntTuple_get(r, 2),
                      y2 = stIntTuple_get(r, 3);
        Sequence *sX3 = SsX->sliceFcn(SsX, x1, x2 - x1), *sY3 = SsY->sliceFcn(SsY, y1, y2 - y1);
        stList *sub = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
        while (j < stList_length(anchorPairs)) {
            stIntTuple *a = stList_get(anchorPairs, j);
            const int64_t x = stIntTuple_get(a, 0), y = stIntTuple_get(a, 1);
            if (x + y >= x2 + y2) break;
            stList_append(sub, stIntTuple_construct2(x - x1, y - y1));
            j++;
        }
        getPosteriorProbsWithBanding(sM, sub, sX3, sY3, p, raggedL || i > 0, raggedR || i < n - 1, fn, extraArgs);
        if (coordinateCorrectionFn) coordinateCorrectionFn(x1, y1, extraArgs);
        stList_destruct(sub);
        sequence_sequenceDestroy(sX3);
        sequence_sequenceDestroy(sY3);
    }
    stList_destruct(splitPoints);
}
stList *getAlignedPairs(StateMachine *sM, void *cX, void *cY, int64_t lX, int64_t lY, PairwiseAlignmentParameters *p,
                        void *(*getXFcn)(void *, int64_t), void *(*getYFcn)(void *, int64_t),
                        stList *(*getAnchorPairFcn)(void *, void *, PairwiseAlignmentParameters *), bool raggedL,
                        bool raggedR) {
    stList *anchorPairs = getAnchorPairFcn(cX, cY, p);
    Sequence *SsX = sequence_construct2(lX, cX, getXFcn, sequence_sliceNucleotideSequence2);
    Sequence *SsY = sequence_construct2(lY, cY, getYFcn, sequence_sliceNucleotideSequence2);
    stList *pairs = getAlignedPairsUsingAnchors(sM, SsX, SsY, anchorPairs, p, diagonalCalculationPosteriorMatchProbs,
                                                raggedL, raggedR);
    sequence_sequenceDestroy(SsX);
    sequence_sequenceDestroy(SsY);
    stList_destruct(anchorPairs);
    return pairs;
}

/* ---- .hmm of the vanilla machine, .expectations reader of the HDP machine (impl/continuousHmm.c:477-900) ----- */
void cpecan_vanillaExpectations_write(VanillaHmmExpectations *hmm, StateMachine *sM, FILE *fh) {
    if (sM->type != vanilla) die("you gave me the wrong type of HMM");
    fprintf(fh, "%i\t%lld\t%lld\t\n", (int) vanilla, 3ll, (long long) NUM_OF_KMERS);
    for (int i = 0; i < 60; i++)
        if (isnan(hmm->kmerSkipBins[i])) {
            fprintf(stdout, "GOT NaN TRANS\n");
            return;
        }
    for (int i = 0; i < 60; i++) fprintf(fh, "%f\t", hmm->kmerSkipBins[i]);
    fprintf(fh, "%f\n", hmm->likelihood);
    const int64_t n = 1 + (int64_t) NUM_OF_KMERS * MODEL_PARAMS;
    for (int64_t i = 0; i < n; i++) fprintf(fh, "%f\t", sM->EMISSION_MATCH_PROBS[i]);
    fprintf(fh, "\n");
    for (int64_t i = 0; i < n; i++) fprintf(fh, "%f\t", sM->EMISSION_GAP_Y_PROBS[i]);
    fprintf(fh, "\n");
}
VanillaHmmExpectations *cpecan_vanillaExpectations_read(const char *fileName, StateMachine *sM) {
    FILE *f = fopen(fileName, "r");
    if (!f) die("cpecan: cannot open %s", fileName);
    double hdr[3], line[61];
    if (line_doubles(f, hdr, 3) != 3 || (int) hdr[0] != (int) vanilla)
        die("Vanilla HMM construct: Wrong HMM type for this function got: %i", (int) hdr[0]);
    if ((int64_t) hdr[2] != NUM_OF_KMERS) die("cpecan: %s: symbol set size %lld", fileName, (long long) hdr[2]);
    const int64_t got = line_doubles(f, line, 61);
    if (got != 61) die("incorrect number of kmer skip bins in HMM %s got %lld instead of 61", fileName, (long long) got);
    VanillaHmmExpectations *hmm = calloc(1, sizeof *hmm);
    memcpy(hmm->kmerSkipBins, line, sizeof(double) * 60);
    hmm->likelihood = line[60];
    const int64_t n = 1 + (int64_t) NUM_OF_KMERS * MODEL_PARAMS;
    double *table = malloc(sizeof(double) * (size_t) n);
    for (int which = 0; which < 2; which++) {
        const int64_t m = line_doubles(f, table, n);
        if (m != n)
            die("incorrect number of members for %s in HMM %s got %lld instead of %lld",
                which ? "extra event match model" : "match model", fileName, (long long) m, (long long) n);
        if (sM) memcpy(which ? sM->EMISSION_GAP_Y_PROBS : sM->EMISSION_MATCH_PROBS, table, sizeof(double) * (size_t) n);
    }
    free(table);
    fclose(f);
    return hmm;
}
HdpHmmExpectations *cpecan_hdpExpectations_read(const char *fileName) {
    FILE *f = fopen(fileName, "r");
    if (!f) die("cpecan: cannot open %s", fileName);
    double hdr[4], line[10];
    const int64_t nh = line_doubles(f, hdr, 4);
    if (nh != 4) die("ERROR loading hdpHmm, got %lld tokens should get 4", (long long) nh);
    if ((int64_t) hdr[1] != 3) die("cpecan: %s: %lld states", fileName, (long long) hdr[1]);
    HdpHmmExpectations *hmm = cpecan_hdpExpectations_construct(0.0, hdr[2]);
    const int64_t n = (int64_t) hdr[3];
    const int64_t nt = line_doubles(f, line, 10);
    if (nt != 10)
        die("Incorrect number of transitions in the input HMM file %s, got %lld instead of %lld", fileName,
            (long long) nt, 10ll);
    memcpy(hmm->transitions, line, sizeof(double) * 9);
    hmm->likelihood = line[9];
    hmm->capacity = n > 0 ? n : 1;
    hmm->eventAssignments = malloc(sizeof(double) * (size_t) hmm->capacity);
    hmm->kmerAssignments = calloc((size_t) hmm->capacity, KMER_LENGTH + 1);
    const int64_t ne = line_doubles(f, hmm->eventAssignments, n);
    if (ne != n) die("Incorrect number of events got %lld, should be %lld", (long long) ne, (long long) n);
    char *l = read_line(f), *p = l;
    int64_t nk = 0;
    while (p && *p) { /* tab-separated k-mers */
        while (*p == '\t' || *p == ' ') p++;
        if (!*p) break;
        const char *start = p;
        while (*p && *p != '\t' && *p != ' ') p++;
        if (p - start != KMER_LENGTH) die("cpecan: %s: k-mer of length %lld", fileName, (long long) (p - start));
        if (nk < n) memcpy(hmm->kmerAssignments + nk * (KMER_LENGTH + 1), start, KMER_LENGTH);
        nk++;
    }
    free(l);
    if (nk != n) die("Incorrect number of kmers got %lld, should be %lld", (long long) nk, (long long) n);
    hmm->numberOfAssignments = n;
    fclose(f);
    return hmm;
}
