/*
 * cpecan_oracle.h -- CPU restatement of cPecan's banded pair-HMM posterior DP.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it.  The shipped path
 * (cpecan-signal_amd/) never links against or falls back to anything in oracle/.
 *
 * Each function names the reference file:line whose arithmetic (operation order,
 * float-suffixed literals, NULL-neighbour handling) it restates.  The reference cannot be
 * built here (it needs sonLib, which is absent; writing a stand-in is not allowed), so the
 * restatement is pinned by the reference's own test vectors -- see tests/test_oracle_*.py.
 */
#ifndef CPECAN_ORACLE_H_
#define CPECAN_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_KMERS 4096
#define ORC_KMER_LEN 6
#define ORC_MODEL_PARAMS 5
#define ORC_PROB_1 10000000 /* PAIR_ALIGNMENT_PROB_1, inc/pairwiseAligner.h:26 */

/* model kinds */
#define ORC_SM3_STRAWMAN 0 /* stateMachine3_cellCalculate + strawMan emissions */
#define ORC_SM5_SYMBOL 1   /* stateMachine5_cellCalculate + symbol emissions   */
#define ORC_SM3_VANILLA 2  /* stateMachine3Vanilla_cellCalculate (impl/stateMachine.c:1368-1409): skip-bin
                              transitions, Gaussian level + inverse-Gaussian noise emissions; X elements are
                              read with sequence_getKmer2.  t[0..4] = TRANSITION_M_TO_Y_NOT_X,
                              TRANSITION_E_TO_E, DEFAULT_END_MATCH_PROB, _FROM_X_PROB, _FROM_Y_PROB;
                              gapX = the 60 skip-bin values (beta[30] | alpha[30]) */

/* sm3 transition slots (log space), order of struct _StateMachine3 inc/stateMachine.h:179-187 */
enum {
    ORC_T3_MATCH_CONTINUE = 0, ORC_T3_MATCH_FROM_GAP_X, ORC_T3_MATCH_FROM_GAP_Y,
    ORC_T3_GAP_OPEN_X, ORC_T3_GAP_OPEN_Y, ORC_T3_GAP_EXTEND_X, ORC_T3_GAP_EXTEND_Y,
    ORC_T3_GAP_SWITCH_TO_X, ORC_T3_GAP_SWITCH_TO_Y, ORC_T3_COUNT
};

#define ORC_SM3_HDP 3      /* stateMachine3HDP_cellCalculate (impl/stateMachine.c:1336-1366): sm3 transitions,
                              X-gap emission log(0.1), match and Y-gap emission = the HDP's posterior-predictive
                              DENSITY at the event mean for the k-mer's Dirichlet process (a linear density used
                              where a log is expected: quirk Q6); X elements read with sequence_getKmer3 */
#define ORC_SM4_SIGNAL 4   /* stateMachine4_cellCalculate (impl/stateMachine.c:867-918): match, short gap X, short gap Y,
                            * long gap X over the strawMan emissions; t[] in the member order of _StateMachine4
                            * (inc/stateMachine.h:134-152): MATCH_CONTINUE, MATCH_FROM_SHORT_GAP_X, MATCH_FROM_LONG_GAP_X,
                            * MATCH_FROM_SHORT_GAP_Y, GAP_SHORT_OPEN_X, GAP_SHORT_EXTEND_X, GAP_SHORT_OPEN_Y,
                            * GAP_SHORT_EXTEND_Y, GAP_LONG_OPEN_X, GAP_LONG_EXTEND_X, GAP_LONG_SWITCH_TO_X */
typedef struct {
    int32_t kind;          /* ORC_SM3_STRAWMAN | ORC_SM5_SYMBOL | ORC_SM3_VANILLA | ORC_SM3_HDP | ORC_SM4_SIGNAL */
    int32_t stateNumber;   /* 3 | 4 | 5 */
    double t[17];          /* transitions: sm3 uses ORC_T3_*, sm5 uses struct order of _StateMachine5 */
    const double *match;   /* sm3: [1+4096*5] EMISSION_MATCH_PROBS; sm5: [16] */
    const double *gapX;    /* sm3: [4096] EMISSION_GAP_X_PROBS;    sm5: [4]  */
    const double *gapY;    /* sm3: [1+4096*5] EMISSION_GAP_Y_PROBS; sm5: [4]  */
    /* HDP (impl/nanopore_hdp.c, impl/hdp.c:2577-2601): per k-mer id (base-alphabetSize, most significant
     * first, :348-380) the row of its nearest observed ancestor's tables; rows of gridLength values */
    const int32_t *hdpRow; /* [alphabetSize^6] */
    const double *hdpGrid; /* [gridLength] sampling grid (linspace, impl/hdp_math_utils.c:497-510) */
    const double *hdpY;    /* [rows][gridLength] posterior predictive */
    const double *hdpSlope;/* [rows][gridLength] spline slopes        */
    int32_t gridLength, alphabetSize;
    char alphabet[16];
} orc_model;

typedef struct {
    double threshold;
    int64_t minDiagsBetweenTraceBack;
    int64_t traceBackDiagonals;
    int64_t diagonalExpansion;
    int64_t splitMatrixBiggerThanThis;
} orc_params;

/* sufficient statistics of the strawMan signal HMM (ContinuousPairHmm, impl/continuousHmm.c:90) */
typedef struct {
    double transitions[9]; /* [from*3+to] */
    double kmerGap[ORC_NUM_KMERS];
    double likelihood;
} orc_expectations;

/* sufficient statistics of the 5-state symbol HMM (HmmDiscrete, impl/discreteHmm.c:10-153 with
 * cell_updateExpectations impl/pairwiseAligner.c:407-424) */
typedef struct {
    double transitions[25]; /* [from*5+to] */
    double emissions[80];   /* [state*16 + x*4 + y] */
    double likelihood;
} orc_expectations5;

/* result container (growable) */
typedef struct {
    int64_t n, cap;
    int64_t *triples;  /* n * 3: (posterior*1e7 floored, x, y) */
    double *logp;      /* n: (F+B)-total exponent that produced each triple (oracle extra) */
    int64_t nTotals, capTotals;
    int64_t *totalsXay;   /* diagonal at which totalProbability was refreshed (global order) */
    double *totals;       /* its value */
    int64_t cells;        /* sum of band widths over all processed sub-alignments */
} orc_result;

void orc_defaults_sm3_nanopore(orc_model *m);   /* impl/stateMachine.c:1278-1289 */
void orc_defaults_sm5(orc_model *m, double *match16, double *gap4x, double *gap4y); /* :60-82,:920-937 */
void orc_defaults_vanilla(orc_model *m);       /* stateMachine3Vanilla_construct :1560-1600 */
void orc_defaults_sm4(orc_model *m);           /* stateMachine4_construct :960-1037 (template-read transitions) */
void orc_defaults_hdp(orc_model *m);           /* stateMachine3Hdp_construct + nanopore defaults */
double orc_grid_spline_interp(double query_x, const double *x, const double *y, const double *slope,
                              int64_t length); /* impl/hdp_math_utils.c:471-495 */
int64_t orc_hdp_kmer_id(const orc_model *m, const char *kmer); /* impl/nanopore_hdp.c:348-380; -1: bad char */
double orc_hdp_density(const orc_model *m, const char *kmer, double x); /* get_nanopore_kmer_density :390 */
double orc_vanilla_match(const double *model, int64_t k, const double *event); /* :499-528 */
void orc_params_default(orc_params *p);         /* impl/pairwiseAligner.c:1428-1441 */

double orc_logAdd(double x, double y);           /* impl/pairwiseAligner.c:238-255 */
int64_t orc_kmer_index(const char *kmer);        /* impl/stateMachine.c:104-139 (6 chars read) */
double orc_logGaussPdf(double x, double mu, double sigma);            /* :333-343 */
double orc_strawman_match(const double *model, int64_t kmerIndex, const double *event); /* :595-629 */
double orc_kmer_gap(const double *gapX, int64_t kmerIndex);           /* :175-187 */
void orc_scale_model(double *match, double scale, double shift, double var, double scale_sd,
                     double var_sd); /* :631-651 */

/* band_construct, impl/pairwiseAligner.c:132-184.  anchors = nAnchors (x,y) pairs.
 * xmyL/xmyR must hold lX+lY+1 entries.  Returns 0, or -1 for an invalid diagonal. */
int orc_band(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY, int64_t expansion,
             int64_t *xmyL, int64_t *xmyR);

/* getSplitPoints, impl/pairwiseAligner.c:1289-1340.  out holds 4-tuples; returns count. */
int64_t orc_split_points(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY,
                         int64_t maxMatrixSize, int raggedLeft, int raggedRight, int64_t *out,
                         int64_t outCap);

orc_result *orc_result_new(void);
void orc_result_free(orc_result *r);

/*
 * getAlignedPairsUsingAnchors (impl/pairwiseAligner.c:1456) with
 * diagonalCalculationPosteriorMatchProbs.  x = char string (sm3: lX k-mers => lX+5 chars;
 * sm5: lX bases), y = events double[3*lY] (sm3) or char string (sm5).
 * If hmm != NULL runs getExpectationsUsingAnchors (:1571) with
 * diagonalCalculation_Expectations instead and emits no triples.
 */
int orc_aligned_pairs_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y,
                                    int64_t lY, const int64_t *anchors, int64_t nAnchors,
                                    const orc_params *p, int raggedLeft, int raggedRight,
                                    orc_expectations *hmm, orc_result *out);

/* getAlignedPairsWithoutBanding, impl/pairwiseAligner.c:1512-1569 */
int orc_aligned_pairs_without_banding(const orc_model *m, const char *x, int64_t lX, const void *y,
                                      int64_t lY, const orc_params *p, int raggedLeft,
                                      int raggedRight, orc_result *out);

/*
 * Debug/verification variant of one getPosteriorProbsWithBanding call (:870): additionally copies
 * every forward diagonal and every backward diagonal (as it stands when its posteriors are taken)
 * into dumpF/dumpB, laid out [sum of widths of earlier diagonals + cell][state].
 */
int orc_banded_dump(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                    const int64_t *anchors, int64_t nAnchors, const orc_params *p, int raggedLeft,
                    int raggedRight, double *dumpF, double *dumpB, orc_result *out);

/* continuousPairHmm_normalize, impl/continuousHmm.c:174-204 */
void orc_expectations_normalize(orc_expectations *e);
/* sufficient statistics of the vanilla signal HMM (VanillaHmm, impl/continuousHmm.c:373-466 with
 * cell_signal_updateBetaAndAlphaProb impl/pairwiseAligner.c:478-498): bins 0..29 beta (match -> gapX),
 * 30..59 alpha (gapX -> gapX) */
typedef struct {
    double kmerSkipBins[60];
    double likelihood;
} orc_expectations_v;
int orc_expectations_v_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                                     const int64_t *anchors, int64_t nAnchors, const orc_params *p,
                                     int raggedLeft, int raggedRight, orc_expectations_v *hmm);
/* sufficient statistics of the HDP signal HMM (HdpHmm, impl/continuousHmm.c:631-697 with
 * cell_signal_updateTransAndKmerSkipExpectations2 impl/pairwiseAligner.c:445-476): transitions, and for every
 * transition into match with posterior >= threshold an assignment of the cell's event to its k-mer, recorded as
 * (from state, X index, Y index) in the order the reference appends them */
typedef struct {
    double transitions[9];
    double likelihood;
    double threshold;
    int64_t n, cap;
    int64_t *assign; /* 3 per assignment */
    double *logp;    /* the transition's log posterior */
} orc_expectations_h;
orc_expectations_h *orc_expectations_h_new(double threshold);
void orc_expectations_h_free(orc_expectations_h *h);
int orc_expectations_h_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                                     const int64_t *anchors, int64_t nAnchors, const orc_params *p,
                                     int raggedLeft, int raggedRight, orc_expectations_h *hmm);
/* getExpectationsUsingAnchors (:1571) for an ORC_SM5_SYMBOL model: adds to *hmm */
int orc_expectations5_using_anchors(const orc_model *m, const char *x, int64_t lX, const void *y, int64_t lY,
                                    const int64_t *anchors, int64_t nAnchors, const orc_params *p,
                                    int raggedLeft, int raggedRight, orc_expectations5 *hmm);
/* hmmDiscrete_normalize2(hmm, TRUE) impl/discreteHmm.c:125-153 */
void orc_expectations5_normalize(orc_expectations5 *e);

#ifdef __cplusplus
}
#endif
#endif
