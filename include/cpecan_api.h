/*
 * cpecan_api.h -- host-side C mirror of the part of cPecan's public API that sits on the banded
 * pair-HMM posterior path (reference: inc/pairwiseAligner.h, inc/stateMachine.h, inc/nanopore.h).
 *
 * Same names, argument meaning and ownership rules as the reference so that its callers
 * (vanillaAlign.c:179-255, the CuTest suites) read unchanged; the DP itself runs on the GPU through
 * the C-ABI of cpecan_hip.h.  The reference's base library sonLib is not part of this build, so the
 * two container types its API returns (stList of stIntTuple) are provided here in minimal form
 * under the reference's names; build with -DCPECAN_WITH_SONLIB to use a real sonLib instead.
 *
 * What is NOT mirrored: function-pointer plug-ins cannot run on the device.  The entry points accept
 * the reference's known combinations -- a threeState (strawMan) StateMachine, sequence_getKmer /
 * sequence_getEvent getters, diagonalCalculationPosteriorMatchProbs -- and abort with a message (the
 * reference's st_errAbort convention) on anything else.  There is no CPU fallback.
 */
#ifndef CPECAN_API_H_
#define CPECAN_API_H_

#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- minimal sonLib containers (inc/sonLibList.h, sonLibTuples.h in sonLib) -------------------- */
#ifndef CPECAN_WITH_SONLIB
typedef struct _stList stList;
typedef struct _stIntTuple stIntTuple;
stList *stList_construct(void);
stList *stList_construct3(int64_t size, void (*destructElement)(void *));
void stList_destruct(stList *list);
int64_t stList_length(stList *list);
void *stList_get(stList *list, int64_t index);
void stList_append(stList *list, void *item);
void stList_sort(stList *list, int (*cmpFn)(const void *a, const void *b)); /* cmpFn gets two elements */
int stIntTuple_cmpFn(const void *a, const void *b);
stIntTuple *stIntTuple_construct2(int64_t a, int64_t b);
stIntTuple *stIntTuple_construct3(int64_t a, int64_t b, int64_t c);
int64_t stIntTuple_get(stIntTuple *t, int64_t index);
int64_t stIntTuple_length(stIntTuple *t);
void stIntTuple_destruct(stIntTuple *t);
#endif

#define PAIR_ALIGNMENT_PROB_1 10000000 /* inc/pairwiseAligner.h:26 */
#define NB_EVENT_PARAMS 3               /* inc/nanopore.h:4 */
#define KMER_LENGTH 6                   /* inc/emissionMatrix.h:4 */
#define NUM_OF_KMERS 4096
#define MODEL_PARAMS 5                  /* inc/stateMachine.h:17 */

/* ---- Sequence (inc/pairwiseAligner.h:29-77) ----------------------------------------------------- */
typedef enum { nucleotide = 0, kmer = 1, event = 2 } SequenceType;
typedef struct _sequence Sequence;
struct _sequence {
    int64_t length;
    void *elements;
    void *(*get)(void *elements, int64_t index);
    Sequence *(*sliceFcn)(Sequence *, int64_t, int64_t);
};
Sequence *sequence_construct(int64_t length, void *elements, void *(*getFcn)(void *, int64_t));
Sequence *sequence_construct2(int64_t length, void *elements, void *(*getFcn)(void *, int64_t),
                              Sequence *(*sliceFcn)(Sequence *, int64_t, int64_t));
Sequence *sequence_sliceNucleotideSequence2(Sequence *inputSequence, int64_t start, int64_t sliceLength);
Sequence *sequence_sliceEventSequence2(Sequence *inputSequence, int64_t start, int64_t sliceLength);
void sequence_sequenceDestroy(Sequence *seq);
void *sequence_getKmer(void *elements, int64_t index);
void *sequence_getBase(void *elements, int64_t index); /* :308-312 */
void *sequence_getKmer2(void *elements, int64_t index); /* :320-325: previous + current k-mer */
void *sequence_getKmer3(void *elements, int64_t index); /* :327-331: index < 0 reads the first k-mer */
Sequence *sequence_sliceNucleotideSequence(Sequence *inputSequence, int64_t start, int64_t sliceLength);
void *sequence_getEvent(void *elements, int64_t index);
int64_t sequence_correctSeqLength(int64_t length, SequenceType type);

/* ---- PairwiseAlignmentParameters (inc/pairwiseAligner.h:80-95) ---------------------------------- */
typedef struct _pairwiseAlignmentBandingParameters {
    double threshold;
    int64_t minDiagsBetweenTraceBack;
    int64_t traceBackDiagonals;
    int64_t diagonalExpansion;
    int64_t constraintDiagonalTrim;
    int64_t anchorMatrixBiggerThanThis;
    int64_t repeatMaskMatrixBiggerThanThis;
    int64_t splitMatrixBiggerThanThis;
    bool alignAmbiguityCharacters;
    float gapGamma;
} PairwiseAlignmentParameters;
PairwiseAlignmentParameters *pairwiseAlignmentBandingParameters_construct(void);
void pairwiseAlignmentBandingParameters_destruct(PairwiseAlignmentParameters *p);

/* ---- StateMachine (inc/stateMachine.h:20-102, 174-195) ------------------------------------------ */
typedef enum {
    fiveState = 0, fiveStateAsymmetric = 1, threeState = 2, threeStateAsymmetric = 3, vanilla = 4,
    echelon = 5, fourState = 6, threeStateHdp = 7,
} StateMachineType;
typedef enum { match = 0, shortGapX = 1, shortGapY = 2, longGapX = 3, longGapY = 4 } State;

typedef struct _stateMachine StateMachine;
/* The reference's "class" layout (inc/stateMachine.h:76-100): callers read the data members directly and may call
 * through the pointers, so both sit where the reference's headers put them (checked by the _Static_asserts at the
 * end of this header).  The pointers hold plain host C (cpecan_internals.c): they serve the reference's tested
 * internals (cell_calculateForward, diagonalCalculation* ...) and callers of their own; the aligner entry points
 * never go through them -- they recognise the known machines and run the GPU kernels. */
typedef void (*DoTransitionFn)(double *fromCells, double *toCells, int64_t from, int64_t to, double eP, double tP,
                               void *extraArgs);
struct _stateMachine {
    StateMachineType type;
    int64_t stateNumber;
    int64_t matchState;
    int64_t parameterSetSize;
    double *EMISSION_MATCH_PROBS;
    double *EMISSION_GAP_X_PROBS;
    double *EMISSION_GAP_Y_PROBS;
    double (*startStateProb)(StateMachine *sM, int64_t state);
    double (*endStateProb)(StateMachine *sM, int64_t state);
    double (*raggedEndStateProb)(StateMachine *sM, int64_t state);
    double (*raggedStartStateProb)(StateMachine *sM, int64_t state);
    void (*cellCalculate)(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                          void *cY, DoTransitionFn doTransition, void *extraArgs);
    void (*cellCalculateUpdateExpectations)(double *fromCells, double *toCells, int64_t from, int64_t to, double eP,
                                            double tP, void *extraArgs);
};
typedef struct _StateMachine3 {
    StateMachine model;
    double TRANSITION_MATCH_CONTINUE;
    double TRANSITION_MATCH_FROM_GAP_X;
    double TRANSITION_MATCH_FROM_GAP_Y;
    double TRANSITION_GAP_OPEN_X;
    double TRANSITION_GAP_OPEN_Y;
    double TRANSITION_GAP_EXTEND_X;
    double TRANSITION_GAP_EXTEND_Y;
    double TRANSITION_GAP_SWITCH_TO_X;
    double TRANSITION_GAP_SWITCH_TO_Y;
    double (*getXGapProbFcn)(const double *emissionXGapProbs, void *i);
    double (*getYGapProbFcn)(const double *emissionYGapProbs, void *x, void *y);
    double (*getMatchProbFcn)(const double *emissionMatchProbs, void *x, void *y);
} StateMachine3;

/* 4-state signal machine (inc/stateMachine.h:132-165): match, short gap X, short gap Y and a long gap X over the strawMan
 * emissions; stateMachine4_cellCalculate impl/stateMachine.c:867-897 */
typedef struct _StateMachine4 {
    StateMachine model;
    double TRANSITION_MATCH_CONTINUE;
    double TRANSITION_MATCH_FROM_SHORT_GAP_X;
    double TRANSITION_MATCH_FROM_LONG_GAP_X;
    double TRANSITION_MATCH_FROM_SHORT_GAP_Y;
    double TRANSITION_GAP_SHORT_OPEN_X;
    double TRANSITION_GAP_SHORT_EXTEND_X;
    double TRANSITION_GAP_SHORT_OPEN_Y;
    double TRANSITION_GAP_SHORT_EXTEND_Y;
    double TRANSITION_GAP_LONG_OPEN_X;
    double TRANSITION_GAP_LONG_EXTEND_X;
    double TRANSITION_GAP_LONG_SWITCH_TO_X;
    double (*getXGapProbFcn)(const double *emissionXGapProbs, void *kmer);
    double (*getYGapProbFcn)(const double *scaledMatchModel, void *kmer, void *event);
    double (*getMatchProbFcn)(const double *matchModel, void *kmer, void *event);
} StateMachine4;

/* 5-state symbol machine of DNA-against-DNA alignment (inc/stateMachine.h:104-124; data members) */
typedef struct _StateMachine5 {
    StateMachine model;
    double TRANSITION_MATCH_CONTINUE;
    double TRANSITION_MATCH_FROM_SHORT_GAP_X;
    double TRANSITION_MATCH_FROM_LONG_GAP_X;
    double TRANSITION_GAP_SHORT_OPEN_X;
    double TRANSITION_GAP_SHORT_EXTEND_X;
    double TRANSITION_GAP_SHORT_SWITCH_TO_X;
    double TRANSITION_GAP_LONG_OPEN_X;
    double TRANSITION_GAP_LONG_EXTEND_X;
    double TRANSITION_GAP_LONG_SWITCH_TO_X;
    double TRANSITION_MATCH_FROM_SHORT_GAP_Y;
    double TRANSITION_MATCH_FROM_LONG_GAP_Y;
    double TRANSITION_GAP_SHORT_OPEN_Y;
    double TRANSITION_GAP_SHORT_EXTEND_Y;
    double TRANSITION_GAP_SHORT_SWITCH_TO_Y;
    double TRANSITION_GAP_LONG_OPEN_Y;
    double TRANSITION_GAP_LONG_EXTEND_Y;
    double TRANSITION_GAP_LONG_SWITCH_TO_Y;
    double (*getXGapProbFcn)(const double *emissionXGapProbs, void *i);
    double (*getYGapProbFcn)(const double *emissionYGapProbs, void *i);
    double (*getMatchProbFcn)(const double *emissionMatchProbs, void *x, void *y);
} StateMachine5;
#define SYMBOL_NUMBER_NO_N 4 /* inc/emissionMatrix.h */
/* stateMachine5_construct (impl/stateMachine.c:896-965), the reference's own signature.  The emission
 * initialiser is called on the host; the aligner's GPU path needs the three probability getters to be the symbol
 * getters below (they name the emission model the device code implements). */
StateMachine *stateMachine5_construct(StateMachineType type, int64_t parameterSetSize,
                                      void (*setEmissionsDefaults)(StateMachine *sM),
                                      double (*gapXProbFcn)(const double *, void *),
                                      double (*gapYProbFcn)(const double *, void *),
                                      double (*matchProbFcn)(const double *, void *, void *),
                                      void (*cellCalcUpdateExpFcn)(double *fromCells, double *toCells,
                                                                   int64_t from, int64_t to, double eP,
                                                                   double tP, void *extraArgs));
void emissions_symbol_setEmissionsToDefaults(StateMachine *sM);                            /* :60-82   */
double emissions_symbol_getGapProb(const double *emissionGapProbs, void *base);            /* :155-163 */
double emissions_symbol_getMatchProb(const double *emissionMatchProbs, void *x, void *y);  /* :165-173 */
void cell_updateExpectations(double *fromCells, double *toCells, int64_t from, int64_t to, double eP,
                             double tP, void *extraArgs); /* impl/pairwiseAligner.c:407-424 (host) */

/* 3-state vanilla signal machine (inc/stateMachine.h:219-231; data members) */
typedef enum _strand { template = 0, complement = 1 } Strand;
typedef struct _StateMachine3vanilla {
    StateMachine model;
    double TRANSITION_M_TO_Y_NOT_X;
    double TRANSITION_E_TO_E;
    double DEFAULT_END_MATCH_PROB;
    double DEFAULT_END_FROM_X_PROB;
    double DEFAULT_END_FROM_Y_PROB;
    double (*getKmerSkipProb)(StateMachine *sM, void *kmerList, bool getAlpha);
    double (*getScaledMatchProbFcn)(const double *scaledEventModel, void *kmer, void *event);
    double (*getMatchProbFcn)(const double *eventModel, void *kmer, void *event);
} StateMachine3Vanilla;
/* getSignalStateMachine3Vanilla (impl/stateMachine.c:1761): tables from a 3-line .model file, the 30
 * skip bins of its second line stored as beta and alpha (:284-297) */
StateMachine *getSignalStateMachine3Vanilla(const char *modelFile);
void stateMachine3Vanilla_setStrandTransitionsToDefaults(StateMachine *sM, Strand strand); /* :1291 */

/* ---- nanopore reads (inc/nanopore.h; SURVEY section 8f N1: the data format either side of the path) ------- */
typedef struct _nanoporeReadAdjustmentParameters {
    double scale, shift, var, scale_sd, var_sd;
} NanoporeReadAdjustmentParameters;
typedef struct _nanoporeRead {
    int64_t readLength;         /* 2D read length in nucleotides */
    int64_t nbTemplateEvents;
    int64_t nbComplementEvents;
    NanoporeReadAdjustmentParameters templateParams;
    NanoporeReadAdjustmentParameters complementParams;
    char *twoDread;
    int64_t *templateEventMap;  /* [readLength] */
    double *templateEvents;     /* [nbTemplateEvents * NB_EVENT_PARAMS]: mean, noise, duration */
    int64_t *complementEventMap;
    double *complementEvents;
    bool scaled;
} NanoporeRead;
/* the 6-line .npRead text format (impl/nanopore.c:40-200): 13 header tokens, the 2D read, template event
 * map, template events, complement event map, complement events */
NanoporeRead *nanopore_loadNanoporeReadFromFile(const char *nanoporeReadFile);
stList *nanopore_remapAnchorPairs(stList *anchorPairs, int64_t *eventMap);                       /* :202 */
stList *nanopore_remapAnchorPairsWithOffset(stList *unmappedPairs, int64_t *eventMap, int64_t mapOffset);
void nanopore_descaleNanoporeRead(NanoporeRead *npRead); /* :228, with the reference's stride (quirk Q5) */
void nanopore_nanoporeReadDestruct(NanoporeRead *npRead);

/* ---- HDP signal machine (inc/nanopore_hdp.h, inc/stateMachine.h:197-216) ------------------------------
 * NanoporeHDP is opaque, as in the reference.  deserialize_nhdp reads a file written by the reference's
 * serialize_nhdp (impl/nanopore_hdp.c:820-905, impl/hdp.c:2880-3273) as far as density queries need it:
 * alphabet, sampling grid, the tree of Dirichlet processes and the finalized distributions of the observed
 * ones.  It must hold a finalized HDP with data (first two flags 1), which is what the reference's
 * aligner loads; building or sampling HDPs is not part of this path. */
typedef struct _nanopore_hdp NanoporeHDP;
NanoporeHDP *deserialize_nhdp(const char *filepath);
void destroy_nanopore_hdp(NanoporeHDP *nhdp);
int64_t get_nanopore_hdp_alphabet_size(NanoporeHDP *nhdp);
char *get_nanopore_hdp_alphabet(NanoporeHDP *nhdp); /* a copy, as in the reference; caller frees */
double get_nanopore_kmer_density(NanoporeHDP *nhdp, void *kmer, void *x); /* impl/nanopore_hdp.c:390 (host; the aligner evaluates densities on the device) */
typedef struct _StateMachine3_HDP {
    StateMachine model;
    double TRANSITION_MATCH_CONTINUE;
    double TRANSITION_MATCH_FROM_GAP_X;
    double TRANSITION_MATCH_FROM_GAP_Y;
    double TRANSITION_GAP_OPEN_X;
    double TRANSITION_GAP_OPEN_Y;
    double TRANSITION_GAP_EXTEND_X;
    double TRANSITION_GAP_EXTEND_Y;
    double TRANSITION_GAP_SWITCH_TO_X;
    double TRANSITION_GAP_SWITCH_TO_Y;
    double (*getXGapProbFcn)(const double *emissionXGapProbs, void *i);
    NanoporeHDP *hdpModel;
    double (*getYGapProbFcn)(NanoporeHDP *hdp, void *x, void *y);
    double (*getMatchProbFcn)(NanoporeHDP *hdp, void *x, void *y);
} StateMachine3_HDP;
StateMachine *getHdpStateMachine3(NanoporeHDP *hdp); /* impl/stateMachine.c:1738 */
/* for callers that batch HDP alignments through the C-ABI themselves: fills a cpecan_hdp_model (cpecan_hip.h) with the
 * machine's transitions and pointers into its NanoporeHDP's tables (valid while the NanoporeHDP lives) */
void cpecan_hdp_machine_as_model(StateMachine *sM, void *cpecan_hdp_model_out);

/* getStrawManStateMachine3 (impl/stateMachine.c:1725): 3-state machine with nanopore default
 * transitions (:1278), log(0.1) k-mer gap table (:1506), emission tables from a 3-line .model file */
StateMachine *getStrawManStateMachine3(const char *modelFile);
/* getStateMachine4 (impl/stateMachine.c:1750-1759): the 4-state machine with its template-read transitions
 * (stateMachine4_construct :992-1011), the pore model's match and extra-event tables, the k-mer gap table left at zero
 * (emissions_signal_initEmissionsToZero).  Posterior decode on the GPU (cpecan_k_general4); the reference has no Hmm
 * container for it. */
StateMachine *getStateMachine4(const char *modelFile);
void stateMachine3_setTransitionsToNanoporeDefaults(StateMachine *sM);
void emissions_signal_scaleModel(StateMachine *sM, double scale, double shift, double var,
                                 double scale_sd, double var_sd); /* :631-651 */
int64_t emissions_discrete_getKmerIndex(void *kmer);              /* :120-139 */
int64_t emissions_discrete_getKmerIndexFromKmer(void *kmer);      /* :141-153: the same over a copy of the 6 characters */
void stateMachine_destruct(StateMachine *stateMachine);

/* ---- the path (inc/pairwiseAligner.h:249-311) ---------------------------------------------------- */
typedef struct _dpMatrix DpMatrix; /* opaque, as in the reference (host internals below) */
typedef void (*DiagonalPosteriorProbFn)(StateMachine *, int64_t, DpMatrix *, DpMatrix *, Sequence *,
                                        Sequence *, double, PairwiseAlignmentParameters *, void *);
/* pass this as diagonalPosteriorProbFn, exactly as the reference's callers do: the aligner entry points recognise
 * it and decode posteriors on the GPU; called directly it is the reference's host function over DpMatrix objects */
void diagonalCalculationPosteriorMatchProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                            DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY,
                                            double totalProbability, PairwiseAlignmentParameters *p,
                                            void *extraArgs);

stList *getAlignedPairsUsingAnchors(StateMachine *sM, Sequence *SsX, Sequence *SsY, stList *anchorPairs,
                                    PairwiseAlignmentParameters *p,
                                    DiagonalPosteriorProbFn diagonalPosteriorProbFn,
                                    bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);

stList *getAlignedPairsWithoutBanding(StateMachine *sM, void *cX, void *cY, int64_t lX, int64_t lY,
                                      PairwiseAlignmentParameters *p,
                                      void *(*getXFcn)(void *, int64_t), void *(*getYFcn)(void *, int64_t),
                                      DiagonalPosteriorProbFn diagonalPosteriorProbFn,
                                      bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);

stList *getSplitPoints(stList *anchorPairs, int64_t lX, int64_t lY, int64_t maxMatrixSize,
                       bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);

/* one getPosteriorProbsWithBanding call (inc/pairwiseAligner.h:269): no splitting; the aligned triples are
 * appended to the stList in ((void **) extraArgs)[0] in the reference's emission order (windows forward,
 * diagonals descending inside a window), as diagonalCalculationPosteriorMatchProbs does */
void getPosteriorProbsWithBanding(StateMachine *sM, stList *anchorPairs, Sequence *sX, Sequence *sY,
                                  PairwiseAlignmentParameters *p, bool alignmentHasRaggedLeftEnd,
                                  bool alignmentHasRaggedRightEnd,
                                  DiagonalPosteriorProbFn diagonalPosteriorProbFn, void *extraArgs);
/* the split driver itself (:331, impl :1356-1422): one getPosteriorProbsWithBanding per sub-region of
 * getSplitPoints, anchors re-based, ragged flags (left || i > 0, right || i < last), then
 * coordinateCorrectionFn(x1, y1, extraArgs) if given.  diagonalPosteriorProbFn is one of the two markers
 * (posterior decode: extraArgs[0] is the stList the triples go to; expectations: extraArgs is the Hmm). */
void getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps(
    StateMachine *sM, stList *anchorPairs, Sequence *SsX, Sequence *SsY, PairwiseAlignmentParameters *p,
    bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd, DiagonalPosteriorProbFn diagonalPosteriorProbFn,
    void (*coordinateCorrectionFn)(int64_t offsetX, int64_t offsetY, void *extraArgs), void *extraArgs);
/* getAlignedPairs (:100, impl :1486-1510): anchors from the caller's function (the reference passes its lastz
 * wrapper), nucleotide-sliced Sequences, getAlignedPairsUsingAnchors; destroys the anchor list it obtained */
stList *getAlignedPairs(StateMachine *sM, void *cX, void *cY, int64_t lX, int64_t lY, PairwiseAlignmentParameters *p,
                        void *(*getXFcn)(void *, int64_t), void *(*getYFcn)(void *, int64_t),
                        stList *(*getAnchorPairFcn)(void *, void *, PairwiseAlignmentParameters *),
                        bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);
/* filterToRemoveOverlap (:324, impl :1160-1200): from (x, y) pairs sorted by x then y, the pairs that are
 * smaller in both coordinates than every later pair and larger in both than every earlier one */
stList *filterToRemoveOverlap(stList *sortedOverlappingPairs);

/* ---- geometry the reference exports and tests (inc/pairwiseAligner.h:139-188; host integer code) -------- */
typedef struct _diagonal {
    int64_t xay;  /* x + y */
    int64_t xmyL; /* smallest x - y */
    int64_t xmyR; /* largest x - y  */
} Diagonal;
/* invalid coordinates (parity of xay + xmy, xmyL > xmyR) end the program with a message: the reference
 * throws a sonLib exception here, which has no counterpart without sonLib */
Diagonal diagonal_construct(int64_t xay, int64_t xmyL, int64_t xmyR);
int64_t diagonal_getXay(Diagonal diagonal);
int64_t diagonal_getMinXmy(Diagonal diagonal);
int64_t diagonal_getMaxXmy(Diagonal diagonal);
int64_t diagonal_getWidth(Diagonal diagonal);
int64_t diagonal_getXCoordinate(int64_t xay, int64_t xmy);
int64_t diagonal_getYCoordinate(int64_t xay, int64_t xmy);
int64_t diagonal_equals(Diagonal diagonal1, Diagonal diagonal2);
typedef struct _band Band;
Band *band_construct(stList *anchorPairs, int64_t lX, int64_t lY, int64_t expansion);
void band_destruct(Band *band);
typedef struct _bandIterator BandIterator;
BandIterator *bandIterator_construct(Band *band);
void bandIterator_destruct(BandIterator *bandIterator);
BandIterator *bandIterator_clone(BandIterator *bandIterator);
Diagonal bandIterator_getNext(BandIterator *bandIterator);
Diagonal bandIterator_getPrevious(BandIterator *bandIterator);
#define LOG_ZERO (-INFINITY)
double logAdd(double x, double y); /* :235-255, host double arithmetic */

/* ---- flat sufficient statistics of the signal machines (additive: what the GPU E-step produces and the batch /
 * distributed drivers reduce; the reference-shaped Hmm subclasses further down are filled from these) ------- */
typedef struct _continuousPairHmmExpectations {
    double likelihood;
    double transitions[9];              /* [from * 3 + to] */
    double individualKmerGapProbs[NUM_OF_KMERS];
} ContinuousPairHmmExpectations;
/* getExpectationsUsingAnchors (:1571) with diagonalCalculation_Expectations for the strawMan model:
 * adds this alignment's expectations to *hmm */
void cpecan_getSignalExpectationsUsingAnchors(StateMachine *sM, ContinuousPairHmmExpectations *hmm,
                                       Sequence *SsX, Sequence *SsY, stList *anchorPairs,
                                       PairwiseAlignmentParameters *p, bool alignmentHasRaggedLeftEnd,
                                       bool alignmentHasRaggedRightEnd);
void cpecan_pairHmmExpectations_normalize(ContinuousPairHmmExpectations *hmm);                 /* :174-191 */
void cpecan_pairHmmExpectations_load(StateMachine *sM, ContinuousPairHmmExpectations *hmm); /* :206-232 */

void cpecan_pairHmmExpectations_write(ContinuousPairHmmExpectations *hmm, FILE *fileHandle);   /* :234-272 */
ContinuousPairHmmExpectations *cpecan_pairHmmExpectations_read(const char *fileName);         /* :274-370 */

/* ---- expectations of the vanilla machine (VanillaHmm, impl/continuousHmm.c:373-466) ------------------- */
typedef struct _vanillaHmmExpectations {
    double likelihood;
    double kmerSkipBins[60]; /* 0..29 beta (match -> gapX), 30..59 alpha (gapX -> gapX) */
} VanillaHmmExpectations;
/* getExpectationsUsingAnchors (:1571) with diagonalCalculation_Expectations for a StateMachine3Vanilla
 * (cell_signal_updateBetaAndAlphaProb :478-498): adds this alignment's expectations to *hmm */
void cpecan_getVanillaExpectationsUsingAnchors(StateMachine *sM, VanillaHmmExpectations *hmm, Sequence *SsX,
                                        Sequence *SsY, stList *anchorPairs, PairwiseAlignmentParameters *p,
                                        bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);
void cpecan_vanillaExpectations_normalize(VanillaHmmExpectations *hmm);                        /* :420-429 */
void cpecan_vanillaExpectations_load(StateMachine *sM, VanillaHmmExpectations *hmm); /* :452-462 */
/* the vanilla .hmm file (:477-626): header, 60 skip bins + likelihood, then the match table and the extra-event
 * table of the state machine the expectations were taken with (vanillaHmm_implantMatchModelsintoHmm :431-443).
 * The loader fills *hmm and, if sM is given, copies the two tables into it. */
void cpecan_vanillaExpectations_write(VanillaHmmExpectations *hmm, StateMachine *sM, FILE *fileHandle);
VanillaHmmExpectations *cpecan_vanillaExpectations_read(const char *fileName, StateMachine *sM);

/* ---- expectations of the HDP machine (HdpHmm, inc/continuousHmm.h:26-36, impl/continuousHmm.c:631-790) ----- */
typedef struct _hdpHmmExpectations {
    double likelihood;
    double transitions[9];     /* [from * 3 + to] */
    double threshold;          /* a transition into match with posterior >= threshold assigns event to k-mer */
    int64_t numberOfAssignments;
    int64_t capacity;
    double *eventAssignments;  /* [numberOfAssignments] event means */
    char *kmerAssignments;     /* [numberOfAssignments][KMER_LENGTH + 1], NUL-terminated copies */
    int64_t *assignmentXY;     /* [numberOfAssignments][3]: where the k-mer and the event sit in the caller's SsX / SsY
                                  and which read of the call they belong to (E-step only; NULL on an object read
                                  from a file) */
} HdpHmmExpectations;
HdpHmmExpectations *cpecan_hdpExpectations_construct(double pseudocount, double threshold);
void cpecan_hdpExpectations_destruct(HdpHmmExpectations *hmm);
/* getExpectationsUsingAnchors (:1571) with diagonalCalculation_Expectations for a StateMachine3_HDP
 * (cell_signal_updateTransAndKmerSkipExpectations2 :445-476): adds this alignment to *hmm */
void cpecan_getHdpExpectationsUsingAnchors(StateMachine *sM, HdpHmmExpectations *hmm, Sequence *SsX, Sequence *SsY,
                                    stList *anchorPairs, PairwiseAlignmentParameters *p,
                                    bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);
void cpecan_hdpExpectations_load(StateMachine *sM, HdpHmmExpectations *hmm);      /* :681-699 */
void cpecan_hdpExpectations_write(HdpHmmExpectations *hmm, FILE *fileHandle);          /* :701-753, the .expectations file */
HdpHmmExpectations *cpecan_hdpExpectations_read(const char *fileName);               /* :755-900, without the HDP update */

/* ---- Hmm / HmmDiscrete: Baum-Welch for the 5-state symbol machine (inc/stateMachine.h:47-74,
 * inc/discreteHmm.h:9-52, impl/discreteHmm.c), the reference's own structs and signatures ---------- */
typedef struct _hmm Hmm;
struct _hmm {
    double likelihood;
    StateMachineType type;
    int64_t stateNumber;
    int64_t symbolSetSize;
    int64_t matrixSize;
    void (*addToTransitionExpectationFcn)(Hmm *hmm, int64_t from, int64_t to, double p);
    void (*setTransitionFcn)(Hmm *hmm, int64_t from, int64_t to, double p);
    double (*getTransitionsExpFcn)(Hmm *hmm, int64_t from, int64_t to);
    void (*addToEmissionExpectationFcn)(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p);
    void (*setEmissionExpectationFcn)(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p);
    double (*getEmissionExpFcn)(Hmm *hmm, int64_t state, int64_t x, int64_t y);
    int64_t (*getElementIndexFcn)(void *);
};
typedef struct _hmmDiscrete {
    Hmm baseHmm;
    double *transitions; /* [from * stateNumber + to] */
    double *emissions;   /* [state * matrixSize + x * symbolSetSize + y] */
} HmmDiscrete;
Hmm *hmmDiscrete_constructEmpty(double pseudocount, int64_t stateNumber, int64_t symbolSetSize,
                                StateMachineType type,
                                void (*addToTransitionExpFcn)(Hmm *hmm, int64_t from, int64_t to, double p),
                                void (*setTransitionFcn)(Hmm *hmm, int64_t from, int64_t to, double p),
                                double (*getTransitionsExpFcn)(Hmm *hmm, int64_t from, int64_t to),
                                void (*addEmissionsExpFcn)(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p),
                                void (*setEmissionExpFcn)(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p),
                                double (*getEmissionExpFcn)(Hmm *hmm, int64_t state, int64_t x, int64_t y),
                                int64_t (*getElementIndexFcn)(void *));
void hmmDiscrete_addToTransitionExpectation(Hmm *hmm, int64_t from, int64_t to, double p);
void hmmDiscrete_setTransitionExpectation(Hmm *hmm, int64_t from, int64_t to, double p);
double hmmDiscrete_getTransitionExpectation(Hmm *hmm, int64_t from, int64_t to);
void hmmDiscrete_addToEmissionExpectation(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p);
void hmmDiscrete_setEmissionExpectation(Hmm *hmm, int64_t state, int64_t x, int64_t y, double p);
double hmmDiscrete_getEmissionExpectation(Hmm *hmm, int64_t state, int64_t x, int64_t y);
void hmmDiscrete_randomizeTransitions(Hmm *hmm);
void hmmDiscrete_randomizeEmissions(Hmm *hmm);
void hmmDiscrete_randomize(Hmm *hmm);                          /* uniform draws, then normalize2(TRUE) */
void hmmDiscrete_normalize2(Hmm *hmm, bool normalizeEmissions); /* :125-153 */
/* declared by the reference (inc/discreteHmm.h:38) and defined nowhere in it; here: transitions and emissions, what
 * hmmDiscrete_randomize (:113-123) calls normalize2 with */
void hmmDiscrete_normalize(Hmm *hmm);
void hmmDiscrete_write(Hmm *hmm, FILE *fileHandle);            /* :156-180: 3 lines, "%f" fields */
Hmm *hmmDiscrete_loadFromFile(const char *fileName);           /* :183-273 */
void hmmDiscrete_destruct(Hmm *hmm);
int64_t emissions_discrete_getBaseIndex(void *base);           /* impl/stateMachine.c:104-118 */
typedef struct _stateMachineFunctions {
    double (*gapXProbFcn)(const double *, void *);
    double (*gapYProbFcn)(const double *, void *);
    double (*matchProbFcn)(const double *, void *, void *);
} StateMachineFunctions;
StateMachineFunctions *stateMachineFunctions_construct(double (*gapXProbFcn)(const double *, void *),
                                                       double (*gapYProbFcn)(const double *, void *),
                                                       double (*matchProbFcn)(const double *, void *, void *));
/* the M-step: a 5-state machine from normalised expectations (impl/stateMachine.c:1698-1723 with
 * stateMachine5_loadSymmetric :1100-1154 for type fiveState, _loadAsymmetric :1051-1098 otherwise) */
StateMachine *getStateMachine5(Hmm *hmmD, StateMachineFunctions *sMfs);
/* like diagonalCalculationPosteriorMatchProbs: names the per-diagonal function of the E-step for the aligner entry
 * points (GPU); called directly it is the reference's host function (:841-863) */
void diagonalCalculation_Expectations(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                      DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY,
                                      double totalProbability, PairwiseAlignmentParameters *p, void *extraArgs);
/* the E-step of one alignment (impl/pairwiseAligner.c:1571-1591): adds to hmmExpectations through its add
 * functions.  Dispatch on sM->type as the reference's callers rely on (vanillaAlign.c:345-356): fiveState /
 * fiveStateAsymmetric with an HmmDiscrete, threeState with a ContinuousPairHmm, vanilla with a VanillaHmm,
 * threeStateHdp with an HdpHmm; every E-step runs on the GPU. */
void getExpectationsUsingAnchors(StateMachine *sM, Hmm *hmmExpectations, Sequence *SsX, Sequence *SsY,
                                 stList *anchorPairs, PairwiseAlignmentParameters *p,
                                 DiagonalPosteriorProbFn diagonalCalcExpectationFcn,
                                 bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);
/* (:1593-1617) with the caller's anchor generator; destroys the anchor list it obtained */
void getExpectations(StateMachine *sM, Hmm *hmmExpectations, void *sX, void *sY, int64_t lX, int64_t lY,
                     PairwiseAlignmentParameters *p, void *(*getFcn)(void *, int64_t),
                     stList *(*getAnchorPairFcn)(void *, void *, PairwiseAlignmentParameters *),
                     bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);

/* ---- re-weighting aligned pairs by the chance of aligning to a gap (impl/pairwiseAligner.c:1619-1667) ----- */
int64_t *getIndelProbabilities(stList *alignedPairs, int64_t seqLength, bool xIfTrueElseY);
stList *reweightAlignedPairs(stList *alignedPairs, int64_t *indelProbsX, int64_t *indelProbsY,
                             double gapGamma); /* consumes alignedPairs */
stList *reweightAlignedPairs2(stList *alignedPairs, int64_t seqLengthX, int64_t seqLengthY, double gapGamma);
void sequence_padSequence(Sequence *sequence); /* :282-285: elements become a padded copy (echelon callers) */

/* ---- aligned pairs -> the 15-column TSV of signalAlign (vanillaAlign.c:26-96) ---------
 * contig, reference position, reference k-mer, read file, strand (t/c), event index, event mean / noise /
 * duration, aligned k-mer, model level mean / noise mean of that k-mer, posterior, de-scaled event mean,
 * de-scaled model mean.  Appends to posteriorProbsFile.  matchModel is the (scaled) match table of the strand's
 * state machine, events the strand's [mean, noise, duration] triples, target the sequence the pairs' x indexes. */
void writePosteriorProbs(char *posteriorProbsFile, char *readFile, double *matchModel, double scale, double shift,
                         double *events, char *target, bool forward, char *contig, int64_t eventSequenceOffset,
                         int64_t referenceSequenceOffset, stList *alignedPairs, Strand strand);

/* ---- additive batch entry (many reads, one call; SURVEY section 8b last row) ---------------------
 * Aligns n reads; read i uses state machine sMs[i] (already scaled for that read), sequences
 * sXs[i] (sequence_getKmer) / sYs[i] (sequence_getEvent) and anchor list anchors[i].  Returns an
 * array of n stLists (caller destructs each, then free()s the array). */
stList **getAlignedPairsUsingAnchorsBatch(int64_t n, StateMachine **sMs, Sequence **sXs, Sequence **sYs,
                                          stList **anchors, PairwiseAlignmentParameters *p,
                                          bool alignmentHasRaggedLeftEnd,
                                          bool alignmentHasRaggedRightEnd);

/* Small exported helpers of the reference that none of its four GPU-backed machines calls (inc/stateMachine.h,
 * inc/emissionMatrix.h, inc/pairwiseAligner.h): plain host functions, same names, signatures and arithmetic. */
void emissions_kmer_setMatchProbsToDefaults(double *emissionMatchProbs);  /* [625] impl/emissionMatrix.c:11 */
void emissions_kmer_setGapProbsToDefaults(double *emissionGapProbs);      /* [25]  impl/emissionMatrix.c:55 */
void emissions_discrete_initEmissionsToZero(StateMachine *sM);            /* impl/stateMachine.c:94 */
double emissions_kmer_getMatchProb(const double *emissionMatchProbs, void *x, void *y);               /* :189 */
double emissions_signal_logGaussMatchProb(const double *eventModel, void *kmer, void *event);         /* :473 */
double emissions_signal_getBivariateGaussPdfMatchProb(const double *eventModel, void *kmer, void *event); /* :556 */
double emissions_signal_getDurationProb(void *event, int64_t n);                                      /* :551 */
double emissions_signal_getKmerSkipProb(StateMachine *sM, void *kmers);                               /* :429 */
void emissions_signal_scaleModelNoiseOnly(StateMachine *sM, double scale, double shift, double var, double scale_sd,
                                          double var_sd);                                             /* :653 */
void stateMachine3_setTransitionsToNucleotideDefaults(StateMachine *sM);                              /* :1265 */
char *diagonal_getString(Diagonal diagonal);                              /* impl/pairwiseAligner.c:81; caller frees */
int sortByXPlusYCoordinate(const void *i, const void *j);                 /* :1011 */
int sortByXPlusYCoordinate2(const void *i, const void *j);                /* :1019 */
void diagonalCalculationMultiPosteriorMatchProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                                 DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY,
                                                 double totalProbability, PairwiseAlignmentParameters *p,
                                                 void *extraArgs);        /* :797 */

/* Anchor generation (impl/pairwiseAligner.c:1065-1281, inc/pairwiseAligner.h:320-322): host code around the external
 * lastz executable, exactly as in the reference ("./cPecanLastz" in the working directory, or the executable the
 * environment variable CPECAN_LASTZ names), read back as exonerate CIGAR lines; match columns trimmed by `trim`,
 * sorted by x + y.  getBlastPairsForPairwiseAlignmentParameters adds the reference's filtering and its second,
 * un-masked pass inside gaps larger than repeatMaskMatrixBiggerThanThis; it is the getAnchorPairFcn the reference
 * hands to getAlignedPairs / getExpectations. */
stList *getBlastPairs(const char *sX, const char *sY, int64_t trim, bool repeatMask);
stList *getBlastPairsForPairwiseAlignmentParameters(void *sX, void *sY, PairwiseAlignmentParameters *p);

/* The E-step of n reads as one batch (getExpectationsUsingAnchors :1571 per read, summed into one Hmm of the
 * machines' type): read i uses sMs[i] as above. */
void getExpectationsUsingAnchorsBatch(int64_t n, StateMachine **sMs, Hmm *hmmExpectations, Sequence **sXs,
                                      Sequence **sYs, stList **anchors, PairwiseAlignmentParameters *p,
                                      bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);
/* The training loop as one native call -- what scripts/trainModels.py:244-330 (signal machines) and
 * cPecanEm.py:107-209 (discrete machine) drive through files and one process per read: `iterations` times an empty
 * Hmm of hmmType with `pseudocount` (hmmContinuous_getEmptyHmm :913 / hmmDiscrete_constructEmpty), the E-step of the
 * nReads reads as one GPU batch, the sum over the ranks of a multi-GPU job (reduce(reduceArg, values, n) sums the
 * vector over all ranks in place; NULL for one rank -- libcpecan_em.so's cpecan_em_comm_reduce does it with one RCCL
 * all-reduce), the normalisation (hmmDiscrete_normalize2 / continuousPairHmm_normalize /
 * vanillaHmm_normalizeKmerSkipBins), and the new parameters loaded into every read's state machine
 * (hmmDiscrete loaders of getStateMachine5 / continuousPairHmm_loadTransitionsAndKmerGapProbs /
 * vanillaHmm_loadKmerSkipBinExpectations / hdpHmm_loadTransitions -- the HDP machine's emission update is the Gibbs
 * sampler, out of scope: its Hmm returns with the event assignments of the last iteration).
 * runningLikelihoods[iteration] receives the summed log-likelihood the iteration started from.  Returns the
 * normalised Hmm of the last iteration (caller destructs it with hmmDiscrete_destruct / hmmContinuous_destruct). */
typedef void (*cpecan_reduce_fn)(void *arg, double *values, int64_t n);
Hmm *cpecan_trainModels(int64_t nReads, StateMachine **sMs, Sequence **sXs, Sequence **sYs, stList **anchorPairs,
                        PairwiseAlignmentParameters *p, bool alignmentHasRaggedLeftEnd,
                        bool alignmentHasRaggedRightEnd, StateMachineType hmmType, int64_t iterations,
                        double pseudocount, double hdpThreshold, cpecan_reduce_fn reduce, void *reduceArg,
                        double *runningLikelihoods);

/* ---- the reference's own signal-EM interface (inc/continuousHmm.h:7-124): Hmm subclasses and their functions,
 * same structs, names and signatures.  getExpectationsUsingAnchors (below, with the 5-state entry) dispatches on
 * sM->type exactly as vanillaAlign.c:345-356 calls it; the E-step runs on the GPU and its sums go through the
 * Hmm's own add functions. */
typedef struct _hmmContinuous {
    Hmm baseHmm;
} HmmContinuous;
typedef struct _strawManHmm {
    HmmContinuous baseContinuousHmm;
    double *transitions;
    double *individualKmerGapProbs;
} ContinuousPairHmm;
typedef struct _vanillaHmm {
    HmmContinuous baseContinuousHmm;
    double *matchModel;
    double *scaledMatchModel;
    double *kmerSkipBins;
    int64_t (*getKmerSkipBin)(double *matchModel, void *cX);
} VanillaHmm;
typedef struct _hdpHmm {
    Hmm baseHmm;
    double *transitions;
    double threshold;
    void (*addToAssignments)(Hmm *, void *, void *);
    stList *eventAssignments; /* double * per assignment */
    stList *kmerAssignments;  /* char * per assignment   */
    int64_t numberOfAssignments;
    NanoporeHDP *nhdp;
} HdpHmm;
Hmm *continuousPairHmm_constructEmpty(
    double pseudocount, int64_t stateNumber, int64_t symbolSetSize, StateMachineType type,
    void (*addToTransitionExpFcn)(Hmm *hmm, int64_t from, int64_t to, double p),
    void (*setTransitionFcn)(Hmm *hmm, int64_t from, int64_t to, double p),
    double (*getTransitionsExpFcn)(Hmm *hmm, int64_t from, int64_t to),
    void (*addToKmerGapExpFcn)(Hmm *hmm, int64_t state, int64_t ki, int64_t ignore, double p),
    void (*setKmerGapExpFcn)(Hmm *hmm, int64_t state, int64_t ki, int64_t ignore, double p),
    double (*getKmerGapExpFcn)(Hmm *hmm, int64_t state, int64_t ki, int64_t ignore),
    int64_t (*getElementIndexFcn)(void *));
void continuousPairHmm_addToTransitionsExpectation(Hmm *hmm, int64_t from, int64_t to, double p);
void continuousPairHmm_setTransitionExpectation(Hmm *hmm, int64_t from, int64_t to, double p);
double continuousPairHmm_getTransitionExpectation(Hmm *hmm, int64_t from, int64_t to);
void continuousPairHmm_addToKmerGapExpectation(Hmm *hmm, int64_t state, int64_t kmerIndex, int64_t ignore, double p);
void continuousPairHmm_setKmerGapExpectation(Hmm *hmm, int64_t state, int64_t kmerIndex, int64_t ignore, double p);
double continuousPairHmm_getKmerGapExpectation(Hmm *hmm, int64_t state, int64_t kmerIndex, int64_t ignore);
void continuousPairHmm_loadTransitionsAndKmerGapProbs(StateMachine *sM, Hmm *hmm); /* :206-232 */
void continuousPairHmm_normalize(Hmm *hmm);                                        /* :174-191 */
void continuousPairHmm_randomize(Hmm *hmm);                                        /* :193-204 */
void continuousPairHmm_destruct(Hmm *hmm);
void continuousPairHmm_writeToFile(Hmm *hmm, FILE *fileHandle);                    /* :234-272 */
Hmm *continuousPairHmm_loadFromFile(const char *fileName);                         /* :274-370 */
Hmm *vanillaHmm_constructEmpty(double pseudocount, int64_t stateNumber, int64_t symbolSetSize, StateMachineType type,
                               void (*addToKmerBinExpFcn)(Hmm *hmm, int64_t bin, int64_t ignore, double p),
                               void (*setKmerBinFcn)(Hmm *hmm, int64_t bin, int64_t ignore, double p),
                               double (*getKmerBinExpFcn)(Hmm *hmm, int64_t bin, int64_t ignore));
void vanillaHmm_addToKmerSkipBinExpectation(Hmm *hmm, int64_t bin, int64_t ignore, double p);
void vanillaHmm_setKmerSkipBinExpectation(Hmm *hmm, int64_t bin, int64_t ignore, double p);
double vanillaHmm_getKmerSkipBinExpectation(Hmm *hmm, int64_t bin, int64_t ignore);
void vanillaHmm_normalizeKmerSkipBins(Hmm *hmm);                                   /* :420-429 */
void vanillaHmm_randomizeKmerSkipBins(Hmm *hmm);
void vanillaHmm_loadKmerSkipBinExpectations(StateMachine *sM, Hmm *hmm);           /* :452-462 */
void vanillaHmm_implantMatchModelsintoHmm(StateMachine *sM, Hmm *hmm);             /* :431-443 */
void vanillaHmm_writeToFile(Hmm *hmm, FILE *fileHandle);                           /* :477-530 */
Hmm *vanillaHmm_loadFromFile(const char *fileName);                                /* :532-626 */
void vanillaHmm_destruct(Hmm *hmm);
Hmm *hdpHmm_constructEmpty(double pseudocount, int64_t stateNumber, StateMachineType type, double threshold,
                           void (*addToTransitionExpFcn)(Hmm *hmm, int64_t from, int64_t to, double p),
                           void (*setTransitionFcn)(Hmm *hmm, int64_t from, int64_t to, double p),
                           double (*getTransitionsExpFcn)(Hmm *hmm, int64_t from, int64_t to));
void hdpHmm_loadTransitions(StateMachine *sM, Hmm *hmm);                           /* :681-699 */
void hdpHmm_writeToFile(Hmm *hmm, FILE *fileHandle);                               /* :701-753 */
/* declared by the reference (inc/continuousHmm.h:109) and defined nowhere in it; here: hdpHmm_loadFromFile */
Hmm *hdpHmm_loadFromFile2(const char *fileName, NanoporeHDP *nHdp);
Hmm *hdpHmm_loadFromFile(const char *fileName, NanoporeHDP *nHdp); /* :755-900: transitions and assignments; the
                                                                       Gibbs update of nHdp is not part of this path */
void hdpHmm_destruct(Hmm *hmm);
void hmmContinuous_loadSignalHmm(const char *hmmFile, StateMachine *sM, StateMachineType type); /* :903-911 */
void hmmContinuous_destruct(Hmm *hmm, StateMachineType type);
Hmm *hmmContinuous_getEmptyHmm(StateMachineType type, double pseudocount, double threshold);    /* :913-945 */
void hmmContinuous_normalize(Hmm *hmm, StateMachineType type);
void hmmContinuous_writeToFile(const char *outFile, Hmm *hmm, StateMachineType type);
int64_t hmmContinuous_howManyAssignments(Hmm *hmm);

/* ---- sonLib's pairwise alignment record (sonLib C/inc/pairwiseAlignment.h, commonC.h: the library the reference links and
 * does not vendor -- include.mk points at a checkout beside it, no version is pinned; the layouts below are sonLib's
 * published ones).  getBlastPairs reads lastz's cigars into these through sonLib's cigarRead; here the cigar lines are
 * parsed directly (cpecan_api.c) and this converter is offered for callers that hold such a record.
 * convertPairwiseForwardStrandAlignmentToAnchorPairs impl/pairwiseAligner.c:1039-1063 ------------------------------ */
#define PAIRWISE_INDEL_X 0
#define PAIRWISE_INDEL_Y 1
#define PAIRWISE_MATCH 2
struct List {
    int64_t length, maxLength;
    void **list;
    void (*destroyElement)(void *);
};
struct AlignmentOperation {
    int64_t opType, length;
    float score;
};
struct PairwiseAlignment {
    char *contig1;
    int64_t start1, end1, strand1;
    char *contig2;
    int64_t start2, end2, strand2;
    float score;
    struct List *operationList;
};
stList *convertPairwiseForwardStrandAlignmentToAnchorPairs(struct PairwiseAlignment *pA, int64_t trim);

/* ---- the constructors that take plug-ins (inc/stateMachine.h:259-312) and the plug-ins themselves ------------ */
StateMachine *stateMachine3_construct(StateMachineType type, int64_t parameterSetSize,
                                      void (*setTransitionsToDefaults)(StateMachine *sM),
                                      void (*setEmissionsDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                      double (*gapXProbFcn)(const double *, void *),
                                      double (*gapYProbFcn)(const double *, void *, void *),
                                      double (*matchProbFcn)(const double *, void *, void *),
                                      void (*cellCalcUpdateExpFcn)(double *fromCells, double *toCells, int64_t from,
                                                                   int64_t to, double eP, double tP, void *extraArgs));
StateMachine *stateMachine4_construct(StateMachineType type, int64_t parameterSetSize,
                                      void (*setEmissionsToDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                      double (*gapXProbFcn)(const double *, void *),
                                      double (*gapYProbFcn)(const double *, void *, void *),
                                      double (*matchProbFcn)(const double *, void *, void *),
                                      void (*cellCalcUpdateFcn)(double *fromCells, double *toCells, int64_t from,
                                                                int64_t to, double eP, double tP, void *extraArgs));
StateMachine *stateMachine3Hdp_construct(StateMachineType type, int64_t parameterSetSize,
                                         void (*setTransitionsToDefaults)(StateMachine *sM),
                                         void (*setEmissionsDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                         NanoporeHDP *hdpModel, double (*gapXProbFcn)(const double *, void *),
                                         double (*gapYProbFcn)(NanoporeHDP *, void *, void *),
                                         double (*matchProbFcn)(NanoporeHDP *, void *, void *),
                                         void (*cellCalcUpdateExpFcn)(double *fromCells, double *toCells, int64_t from,
                                                                      int64_t to, double eP, double tP,
                                                                      void *extraArgs));
StateMachine *stateMachine3Vanilla_construct(StateMachineType type, int64_t parameterSetSize,
                                             void (*setEmissionsDefaults)(StateMachine *sM, int64_t nbSkipParams),
                                             double (*xSkipProbFcn)(StateMachine *, void *, bool),
                                             double (*scaledMatchProbFcn)(const double *, void *, void *),
                                             double (*matchProbFcn)(const double *, void *, void *),
                                             void (*cellCalcUpdateExpFcn)(double *fromCells, double *toCells,
                                                                          int64_t from, int64_t to, double eP,
                                                                          double tP, void *extraArgs));
void emissions_signal_initEmissionsToZero(StateMachine *sM, int64_t nbSkipParams);                 /* :189-219 */
double emissions_kmer_getGapProb(const double *emissionGapProbs, void *kmer);                      /* :175-187 */
double emissions_signal_strawManGetKmerEventMatchProb(const double *eventModel, void *x_i, void *e_j); /* :595-629 */
double emissions_signal_getEventMatchProbWithTwoDists(const double *eventModel, void *kmer, void *event); /* :499-528 */
int64_t emissions_signal_getKmerSkipBin(double *matchModel, void *kmers);                          /* :388-419 */
double emissions_signal_getBetaOrAlphaSkipProb(StateMachine *sM, void *kmers, bool getAlpha);      /* :421-428 */
void cell_signal_updateTransAndKmerSkipExpectations(double *fromCells, double *toCells, int64_t from, int64_t to,
                                                    double eP, double tP, void *extraArgs); /* pairwiseAligner.c:426 */
void cell_signal_updateTransAndKmerSkipExpectations2(double *fromCells, double *toCells, int64_t from, int64_t to,
                                                     double eP, double tP, void *extraArgs); /* :445 */
void cell_signal_updateBetaAndAlphaProb(double *fromCells, double *toCells, int64_t from, int64_t to, double eP,
                                        double tP, void *extraArgs); /* :478 */

/* ---- internals the reference exports and tests (inc/pairwiseAligner.h:190-267): plain host C over DpMatrix
 * objects, for callers and tests that drive the recurrence diagonal by diagonal themselves.  The aligner entry
 * points above do not use them. */
void cell_calculateForward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                           void *cY, void *extraArgs);
void cell_calculateBackward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, void *cX,
                            void *cY, void *extraArgs);
double cell_dotProduct(double *cell1, double *cell2, int64_t stateNumber);
double cell_dotProduct2(double *cell1, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t));
typedef struct _dpDiagonal DpDiagonal;
DpDiagonal *dpDiagonal_construct(Diagonal diagonal, int64_t stateNumber);
DpDiagonal *dpDiagonal_clone(DpDiagonal *diagonal);
bool dpDiagonal_equals(DpDiagonal *diagonal1, DpDiagonal *diagonal2);
void dpDiagonal_destruct(DpDiagonal *dpDiagonal);
double *dpDiagonal_getCell(DpDiagonal *dpDiagonal, int64_t xmy);
double dpDiagonal_dotProduct(DpDiagonal *diagonal1, DpDiagonal *diagonal2);
void dpDiagonal_zeroValues(DpDiagonal *diagonal);
void dpDiagonal_initialiseValues(DpDiagonal *diagonal, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t));
DpMatrix *dpMatrix_construct(int64_t diagonalNumber, int64_t stateNumber);
void dpMatrix_destruct(DpMatrix *dpMatrix);
DpDiagonal *dpMatrix_getDiagonal(DpMatrix *dpMatrix, int64_t xay);
int64_t dpMatrix_getActiveDiagonalNumber(DpMatrix *dpMatrix);
DpDiagonal *dpMatrix_createDiagonal(DpMatrix *dpMatrix, Diagonal diagonal);
void dpMatrix_deleteDiagonal(DpMatrix *dpMatrix, int64_t xay);
void diagonalCalculationForward(StateMachine *sM, int64_t xay, DpMatrix *dpMatrix, Sequence *sX, Sequence *sY);
void diagonalCalculationBackward(StateMachine *sM, int64_t xay, DpMatrix *dpMatrix, Sequence *sX, Sequence *sY);
double diagonalCalculationTotalProbability(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix,
                                           DpMatrix *backwardDpMatrix, Sequence *sX, Sequence *sY);

/* ---- layout checks against the reference's headers (LP64): base struct 104 bytes, the subclasses' first transition
 * at 104, their function pointers behind the transitions ---------------------------------------------------- */
#include <stddef.h>
_Static_assert(offsetof(struct _stateMachine, EMISSION_MATCH_PROBS) == 32, "StateMachine layout");
_Static_assert(offsetof(struct _stateMachine, startStateProb) == 56, "StateMachine layout");
_Static_assert(offsetof(struct _stateMachine, cellCalculate) == 88, "StateMachine layout");
_Static_assert(sizeof(struct _stateMachine) == 104, "StateMachine layout");
_Static_assert(offsetof(StateMachine3, TRANSITION_MATCH_CONTINUE) == 104, "StateMachine3 layout");
_Static_assert(offsetof(StateMachine3, getXGapProbFcn) == 176 && sizeof(StateMachine3) == 200, "StateMachine3 layout");
_Static_assert(offsetof(StateMachine4, TRANSITION_GAP_LONG_SWITCH_TO_X) == 184 && offsetof(StateMachine4, getXGapProbFcn) == 192 &&
               sizeof(StateMachine4) == 216, "StateMachine4 layout");
_Static_assert(offsetof(StateMachine5, getXGapProbFcn) == 240 && sizeof(StateMachine5) == 264, "StateMachine5 layout");
_Static_assert(offsetof(StateMachine3_HDP, hdpModel) == 184 && sizeof(StateMachine3_HDP) == 208, "StateMachine3_HDP layout");
_Static_assert(offsetof(StateMachine3Vanilla, getKmerSkipProb) == 144 && sizeof(StateMachine3Vanilla) == 168,
               "StateMachine3Vanilla layout");
_Static_assert(sizeof(struct _hmm) == 96 && offsetof(HmmDiscrete, transitions) == 96, "Hmm layout");
_Static_assert(offsetof(ContinuousPairHmm, individualKmerGapProbs) == 104, "ContinuousPairHmm layout");
_Static_assert(offsetof(VanillaHmm, getKmerSkipBin) == 120 && offsetof(HdpHmm, nhdp) == 144, "VanillaHmm / HdpHmm layout");
_Static_assert(sizeof(PairwiseAlignmentParameters) == 72, "PairwiseAlignmentParameters layout");

#ifdef __cplusplus
}
#endif
#endif
