/*
 * reference_api_test.c -- a C caller of include/cpecan_api.h written the way the reference's own callers and CuTest
 * suites use the API (tests/pairwiseAlignerTest.c, tests/signalPairwiseTest.c, tests/nanoporeHdpTests.c,
 * vanillaAlign.c): members read straight off the structs, calls through the function pointers, the Hmm subclasses
 * of inc/continuousHmm.h.  Compiled against the header only; links libcpecan_host.so.
 *
 *   reference_api_test cpu <goldenDir>
 *       host-only checks: struct members and function pointers, the cell and diagonal internals with the
 *       reference's known answers, the Hmm containers' file round trips.  Prints one "ok <name>" line per check.
 *   reference_api_test estep <type> <model|nhdp> <target.txt> <events.f64> <anchors.txt> <eventMap.i64>
 *                      <scale> <shift> <var> <scale_sd> <var_sd> <threshold> <out.hmm>
 *       the call sequence of getSignalExpectations (vanillaAlign.c:318-359) for one strand on the GPU; prints the
 *       Hmm's sums at full precision for the Python test to compare with the oracle.
 *   reference_api_test gpu <goldenDir>
 *       the toy alignments of the reference's diagonal tests, once through the exported host internals and once
 *       through the aligner entry points (GPU): same pairs, same integers; expectations agree to 1e-9.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cpecan_api.h"

static int failures = 0;
#define CHECK(cond)                                                                                 \
    do {                                                                                            \
        if (!(cond)) {                                                                              \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                       \
            failures++;                                                                             \
        }                                                                                           \
    } while (0)
#define CLOSE(a, b, tol) CHECK(fabs((a) - (b)) <= (tol))

static char *path_in(const char *dir, const char *name) {
    char *p = malloc(strlen(dir) + strlen(name) + 2);
    sprintf(p, "%s/%s", dir, name);
    return p;
}

/* ---- cells: forward and backward around one cell give the same total (test_strawMan_cell :176, test_vanilla_cell
 * :300, test_cell pairwiseAlignerTest.c:185, test_sm3hdp_cell nanoporeHdpTests.c:495) ------------------------- */
static void cell_round(StateMachine *sM, void *kX, void *eY, double tol, const char *name) {
    double lowerF[5], middleF[5], upperF[5], currentF[5], lowerB[5], middleB[5], upperB[5], currentB[5];
    for (int64_t i = 0; i < sM->stateNumber; i++) {
        middleF[i] = sM->startStateProb(sM, i);
        currentB[i] = sM->endStateProb(sM, i);
        middleB[i] = lowerF[i] = lowerB[i] = upperF[i] = upperB[i] = currentF[i] = LOG_ZERO;
    }
    cell_calculateForward(sM, lowerF, NULL, NULL, middleF, kX, eY, NULL);
    cell_calculateForward(sM, upperF, middleF, NULL, NULL, kX, eY, NULL);
    cell_calculateForward(sM, currentF, lowerF, middleF, upperF, kX, eY, NULL);
    cell_calculateBackward(sM, currentB, lowerB, middleB, upperB, kX, eY, NULL);
    cell_calculateBackward(sM, upperB, middleB, NULL, NULL, kX, eY, NULL);
    cell_calculateBackward(sM, lowerB, NULL, NULL, middleB, kX, eY, NULL);
    const double f = cell_dotProduct2(currentF, sM, sM->endStateProb);
    const double b = cell_dotProduct2(middleB, sM, sM->startStateProb);
    CHECK(isfinite(f));
    CLOSE(f, b, tol);
    printf("%s %s forward %.12g backward %.12g\n", failures ? "FAILED" : "ok", name, f, b);
}

static double toyEvents5[15] = { 60.032615, 0.791316, 0.005, 60.332089, 0.620198, 0.012, 61.618848, 0.747567, 0.008,
                                 66.015805, 0.714290, 0.021, 59.783408, 1.128591, 0.002 };
static double toyEvents7[21] = { 58.743435, 0.887833, 0.0571, 53.604965, 0.816836, 0.0571, 58.432015, 0.735143, 0.0571,
                                 63.684352, 0.795437, 0.0571, 58.921430, 0.812959, 0.0571, 59.895882, 0.740952, 0.0571,
                                 61.684303, 0.722332, 0.0571 };

/* ---- the un-banded toy DP of test_diagonalDPCalculations, driven through the exported internals; returns the
 * aligned pairs at the threshold and the total probability ------------------------------------------------- */
static stList *toy_dp(StateMachine *sM, Sequence *SsX, Sequence *SsY, double threshold, double *total,
                      Hmm *expectations, int64_t expansion, int asBandedAligner) {
    const int64_t lX = SsX->length, lY = SsY->length;
    DpMatrix *F = dpMatrix_construct(lX + lY, sM->stateNumber), *B = dpMatrix_construct(lX + lY, sM->stateNumber);
    stList *anchorPairs = stList_construct();
    Band *band = band_construct(anchorPairs, lX, lY, expansion);
    BandIterator *it = bandIterator_construct(band);
    for (int64_t i = 0; i <= lX + lY; i++) {
        Diagonal d = bandIterator_getNext(it);
        dpDiagonal_zeroValues(dpMatrix_createDiagonal(B, d));
        dpDiagonal_zeroValues(dpMatrix_createDiagonal(F, d));
    }
    CHECK(dpMatrix_getActiveDiagonalNumber(F) == lX + lY + 1);
    dpDiagonal_initialiseValues(dpMatrix_getDiagonal(F, 0), sM, sM->startStateProb);
    dpDiagonal_initialiseValues(dpMatrix_getDiagonal(B, lX + lY), sM, sM->endStateProb);
    for (int64_t i = 1; i <= lX + lY; i++) diagonalCalculationForward(sM, i, F, SsX, SsY);
    for (int64_t i = lX + lY; i > 0; i--) diagonalCalculationBackward(sM, i, B, SsX, SsY);
    const double f = cell_dotProduct2(dpDiagonal_getCell(dpMatrix_getDiagonal(F, lX + lY), lX - lY), sM, sM->endStateProb);
    const double b = cell_dotProduct2(dpDiagonal_getCell(dpMatrix_getDiagonal(B, 0), 0), sM, sM->startStateProb);
    CLOSE(f, b, 0.001);
    for (int64_t i = 0; i <= lX + lY; i++) CLOSE(f, diagonalCalculationTotalProbability(sM, i, F, B, SsX, SsY), 0.01);
    stList *alignedPairs = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    void *extraArgs[1] = { alignedPairs };
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    p->threshold = threshold;
    if (!asBandedAligner) { /* the reference's diagonal tests: one total, diagonals ascending */
        for (int64_t i = 1; i <= lX + lY; i++)
            diagonalCalculationPosteriorMatchProbs(sM, i, F, B, SsX, SsY, f, p, extraArgs);
    } else { /* getPosteriorProbsWithBanding's single traceback (impl/pairwiseAligner.c:934-975): diagonals descending,
                the total refreshed on every tenth */
        double t = LOG_ZERO;
        int64_t done = 0;
        for (int64_t i = lX + lY; i > 0; i--) {
            if (done++ % 10 == 0) t = diagonalCalculationTotalProbability(sM, i, F, B, SsX, SsY);
            diagonalCalculationPosteriorMatchProbs(sM, i, F, B, SsX, SsY, t, p, extraArgs);
            if (expectations) diagonalCalculation_Expectations(sM, i, F, B, SsX, SsY, t, p, expectations);
        }
    }
    pairwiseAlignmentBandingParameters_destruct(p);
    for (int64_t i = 0; i <= lX + lY; i++) {
        dpMatrix_deleteDiagonal(F, i);
        dpMatrix_deleteDiagonal(B, i);
    }
    CHECK(dpMatrix_getActiveDiagonalNumber(F) == 0);
    dpMatrix_destruct(F);
    dpMatrix_destruct(B);
    bandIterator_destruct(it);
    band_destruct(band);
    stList_destruct(anchorPairs);
    *total = f;
    return alignedPairs;
}
static void expect_pairs(stList *got, const int64_t (*want)[2], int n, const char *name) {
    CHECK(stList_length(got) == n);
    for (int64_t i = 0; i < stList_length(got); i++) {
        stIntTuple *t = stList_get(got, i);
        int found = 0;
        for (int k = 0; k < n; k++) found |= stIntTuple_get(t, 1) == want[k][0] && stIntTuple_get(t, 2) == want[k][1];
        CHECK(found);
        CHECK(stIntTuple_get(t, 0) > 0 && stIntTuple_get(t, 0) <= PAIR_ALIGNMENT_PROB_1);
    }
    printf("%s %s %lld pairs\n", failures ? "FAILED" : "ok", name, (long long) stList_length(got));
}

static void test_dp_diagonal_and_matrix(void) { /* test_dpDiagonal :208, test_dpMatrix :243 */
    StateMachine *sM = stateMachine5_construct(fiveState, SYMBOL_NUMBER_NO_N, emissions_symbol_setEmissionsToDefaults,
                                               emissions_symbol_getGapProb, emissions_symbol_getGapProb,
                                               emissions_symbol_getMatchProb, cell_updateExpectations);
    Diagonal diagonal = diagonal_construct(3, -1, 1);
    DpDiagonal *d = dpDiagonal_construct(diagonal, sM->stateNumber);
    double *c1 = dpDiagonal_getCell(d, -1), *c2 = dpDiagonal_getCell(d, 1);
    CHECK(c1 != NULL && c2 != NULL && c2 == c1 + sM->stateNumber);
    CHECK(dpDiagonal_getCell(d, 3) == NULL && dpDiagonal_getCell(d, -3) == NULL);
    dpDiagonal_initialiseValues(d, sM, sM->endStateProb);
    double total = LOG_ZERO;
    for (int64_t i = 0; i < sM->stateNumber; i++) {
        CLOSE(c1[i], sM->endStateProb(sM, i), 0.0);
        CLOSE(c2[i], sM->endStateProb(sM, i), 0.0);
        total = logAdd(total, 2 * c1[i]);
        total = logAdd(total, 2 * c1[i]);
    }
    DpDiagonal *d2 = dpDiagonal_clone(d);
    CHECK(dpDiagonal_equals(d, d2));
    CLOSE(dpDiagonal_dotProduct(d, d2), total, 0.001);
    dpDiagonal_zeroValues(d2);
    CHECK(!dpDiagonal_equals(d, d2));
    CHECK(dpDiagonal_getCell(d2, 1)[0] == LOG_ZERO);
    dpDiagonal_destruct(d);
    dpDiagonal_destruct(d2);

    const int64_t lX = 3, lY = 2;
    DpMatrix *m = dpMatrix_construct(lX + lY, sM->stateNumber);
    CHECK(dpMatrix_getActiveDiagonalNumber(m) == 0);
    for (int64_t i = -1; i <= lX + lY + 10; i++) CHECK(dpMatrix_getDiagonal(m, i) == NULL);
    for (int64_t i = 0; i <= lX + lY; i++) {
        DpDiagonal *x = dpMatrix_createDiagonal(m, diagonal_construct(i, -i, i));
        CHECK(x == dpMatrix_getDiagonal(m, i));
        CHECK(dpMatrix_getActiveDiagonalNumber(m) == i + 1);
    }
    for (int64_t i = lX + lY; i >= 0; i--) {
        dpMatrix_deleteDiagonal(m, i);
        CHECK(dpMatrix_getDiagonal(m, i) == NULL);
        CHECK(dpMatrix_getActiveDiagonalNumber(m) == i);
    }
    dpMatrix_destruct(m);
    stateMachine_destruct(sM);
    printf("%s dpDiagonal_dpMatrix\n", failures ? "FAILED" : "ok");
}

/* ---- the Hmm subclasses' file round trips (test_continuousPairHmm :1461, test_vanillaHmm :1544, test_hdpHmm
 * nanoporeHdpTests.c:905-972) ------------------------------------------------------------------------------ */
static void test_hmm_containers(const char *goldenDir) {
    char tmp[256];
    snprintf(tmp, sizeof tmp, "/tmp/cpecan_reference_api_%ld.hmm", (long) getpid());
    Hmm *hmm = continuousPairHmm_constructEmpty(
        0.0, 3, NUM_OF_KMERS, threeState, continuousPairHmm_addToTransitionsExpectation,
        continuousPairHmm_setTransitionExpectation, continuousPairHmm_getTransitionExpectation,
        continuousPairHmm_addToKmerGapExpectation, continuousPairHmm_setKmerGapExpectation,
        continuousPairHmm_getKmerGapExpectation, emissions_discrete_getKmerIndex);
    ContinuousPairHmm *cp = (ContinuousPairHmm *) hmm;
    const int64_t nStates = cp->baseContinuousHmm.baseHmm.stateNumber, nSymbols = cp->baseContinuousHmm.baseHmm.symbolSetSize;
    for (int64_t from = 0; from < nStates; from++)
        for (int64_t to = 0; to < nStates; to++) hmm->addToTransitionExpectationFcn(hmm, from, to, from * nStates + to);
    double dummyTotal = 0.0;
    for (int64_t i = 0; i < nSymbols; i++) {
        hmm->setEmissionExpectationFcn(hmm, 0, i, 0, nSymbols * nStates + i);
        dummyTotal += nSymbols * nStates + i;
    }
    CHECK(cp->transitions[5] == 5.0 && cp->individualKmerGapProbs[7] == nSymbols * nStates + 7);
    FILE *fH = fopen(tmp, "w");
    continuousPairHmm_writeToFile(hmm, fH);
    fclose(fH);
    continuousPairHmm_destruct(hmm);
    hmm = continuousPairHmm_loadFromFile(tmp);
    CHECK(hmm->type == threeState && hmm->stateNumber == 3 && hmm->symbolSetSize == NUM_OF_KMERS);
    for (int64_t from = 0; from < nStates; from++)
        for (int64_t to = 0; to < nStates; to++) CHECK(hmm->getTransitionsExpFcn(hmm, from, to) == from * nStates + to);
    for (int64_t i = 0; i < nSymbols; i++) CHECK(hmm->getEmissionExpFcn(hmm, 0, i, 0) == nSymbols * nStates + i);
    continuousPairHmm_normalize(hmm);
    for (int64_t from = 0; from < nStates; from++)
        for (int64_t to = 0; to < nStates; to++) {
            const double z = from * nStates * nStates + (nStates * (nStates - 1)) / 2;
            CLOSE(hmm->getTransitionsExpFcn(hmm, from, to), (from * nStates + to) / z, 0.0);
        }
    for (int64_t i = 0; i < nSymbols; i++) CLOSE(hmm->getEmissionExpFcn(hmm, 0, i, 0), (nSymbols * nStates + i) / dummyTotal, 1e-12);
    /* the M-step into a machine, then the loader that goes through a file */
    char *model = path_in(goldenDir, "template_median68pA.model");
    StateMachine *sM = getStrawManStateMachine3(model);
    continuousPairHmm_loadTransitionsAndKmerGapProbs(sM, hmm);
    StateMachine3 *sM3 = (StateMachine3 *) sM;
    CHECK(sM3->TRANSITION_MATCH_CONTINUE == log(hmm->getTransitionsExpFcn(hmm, match, match)));
    CHECK(sM3->TRANSITION_GAP_EXTEND_X == log(1 - hmm->getTransitionsExpFcn(hmm, shortGapX, match)));
    CHECK(sM3->TRANSITION_GAP_SWITCH_TO_Y == LOG_ZERO);
    CHECK(sM->EMISSION_GAP_X_PROBS[11] == log(hmm->getEmissionExpFcn(hmm, 0, 11, 0)));
    hmmContinuous_writeToFile(tmp, hmm, threeState);
    StateMachine *sMb = getStrawManStateMachine3(model);
    hmmContinuous_loadSignalHmm(tmp, sMb, threeState);
    CLOSE(((StateMachine3 *) sMb)->TRANSITION_GAP_OPEN_Y, sM3->TRANSITION_GAP_OPEN_Y, 1e-4); /* "%f" keeps six decimals */
    stateMachine_destruct(sMb);
    stateMachine_destruct(sM);
    continuousPairHmm_destruct(hmm);
    printf("%s continuousPairHmm\n", failures ? "FAILED" : "ok");

    hmm = vanillaHmm_constructEmpty(0.0, 3, NUM_OF_KMERS, vanilla, vanillaHmm_addToKmerSkipBinExpectation,
                                    vanillaHmm_setKmerSkipBinExpectation, vanillaHmm_getKmerSkipBinExpectation);
    dummyTotal = 0.0;
    for (int64_t i = 0; i < 60; i++) {
        hmm->setTransitionFcn(hmm, i, 0, hmm->symbolSetSize * hmm->stateNumber + i);
        dummyTotal += hmm->symbolSetSize * hmm->stateNumber + i;
    }
    StateMachine *sMt = getSignalStateMachine3Vanilla(model);
    vanillaHmm_implantMatchModelsintoHmm(sMt, hmm);
    fH = fopen(tmp, "w");
    vanillaHmm_writeToFile(hmm, fH);
    fclose(fH);
    vanillaHmm_destruct(hmm);
    hmm = vanillaHmm_loadFromFile(tmp);
    VanillaHmm *vHmm = (VanillaHmm *) hmm;
    for (int64_t i = 0; i < 60; i++) CLOSE(hmm->getTransitionsExpFcn(hmm, i, 0), hmm->symbolSetSize * hmm->stateNumber + i, 0.001);
    for (int64_t i = 0; i < 1 + sMt->parameterSetSize * MODEL_PARAMS; i++) {
        CLOSE(vHmm->matchModel[i], sMt->EMISSION_MATCH_PROBS[i], 0.001);
        CLOSE(vHmm->scaledMatchModel[i], sMt->EMISSION_GAP_Y_PROBS[i], 0.001);
    }
    CHECK(vHmm->getKmerSkipBin(vHmm->matchModel, "ACGATAC") == emissions_signal_getKmerSkipBin(sMt->EMISSION_MATCH_PROBS, "ACGATAC"));
    vanillaHmm_normalizeKmerSkipBins(hmm);
    for (int64_t i = 0; i < 60; i++) CLOSE(hmm->getTransitionsExpFcn(hmm, i, 0), (hmm->symbolSetSize * hmm->stateNumber + i) / dummyTotal, 1e-9);
    vanillaHmm_loadKmerSkipBinExpectations(sMt, hmm);
    CHECK(sMt->EMISSION_GAP_X_PROBS[42] == hmm->getTransitionsExpFcn(hmm, 42, 0));
    vanillaHmm_destruct(hmm);
    stateMachine_destruct(sMt);
    printf("%s vanillaHmm\n", failures ? "FAILED" : "ok");

    hmm = hdpHmm_constructEmpty(0.0, 3, threeStateHdp, 0.02, continuousPairHmm_addToTransitionsExpectation,
                                continuousPairHmm_setTransitionExpectation, continuousPairHmm_getTransitionExpectation);
    HdpHmm *hdpHmm = (HdpHmm *) hmm;
    for (int64_t from = 0; from < 3; from++)
        for (int64_t to = 0; to < 3; to++) hmm->addToTransitionExpectationFcn(hmm, from, to, from * 3 + to);
    char *sequence = "ACGTCATACATGACTATA";
    double fakeMeans[3] = { 65.0, 64.0, 63.0 };
    for (int64_t a = 0; a < 3; a++) hdpHmm->addToAssignments(hmm, sequence + a * KMER_LENGTH, fakeMeans + a);
    CHECK(hdpHmm->numberOfAssignments == 3 && hmmContinuous_howManyAssignments(hmm) == 3);
    CHECK(stList_length(hdpHmm->eventAssignments) == 3 && stList_get(hdpHmm->kmerAssignments, 1) == sequence + KMER_LENGTH);
    fH = fopen(tmp, "w");
    hdpHmm_writeToFile(hmm, fH);
    fclose(fH);
    hdpHmm_destruct(hmm);
    hmm = hdpHmm_loadFromFile(tmp, NULL);
    hdpHmm = (HdpHmm *) hmm;
    CHECK(hmm->type == threeStateHdp && hdpHmm->threshold == 0.02 && hdpHmm->numberOfAssignments == 3);
    for (int64_t from = 0; from < 3; from++)
        for (int64_t to = 0; to < 3; to++) CHECK(hmm->getTransitionsExpFcn(hmm, from, to) == from * 3 + to);
    for (int64_t a = 0; a < 3; a++) {
        CHECK(*(double *) stList_get(hdpHmm->eventAssignments, a) == fakeMeans[a]);
        CHECK(strncmp(stList_get(hdpHmm->kmerAssignments, a), sequence + a * KMER_LENGTH, KMER_LENGTH) == 0);
    }
    hmmContinuous_destruct(hmm, threeStateHdp);
    hmm = hmmContinuous_getEmptyHmm(threeStateHdp, 0.001, 0.3);
    CHECK(((HdpHmm *) hmm)->threshold == 0.3 && ((HdpHmm *) hmm)->transitions[8] == 0.001);
    hmmContinuous_destruct(hmm, threeStateHdp);
    remove(tmp);
    free(model);
    printf("%s hdpHmm\n", failures ? "FAILED" : "ok");
}

static Sequence *kmer_sequence(char *chars, void *(*get)(void *, int64_t)) {
    return sequence_construct2(sequence_correctSeqLength((int64_t) strlen(chars), event), chars, get,
                               sequence_sliceNucleotideSequence2);
}

/* the small exported helpers none of the four GPU-backed machines calls (include/cpecan_api.h, "Small exported
 * helpers"): values against the reference's literal tables and formulas */
static void test_small_helpers(const char *model) {
    const double M = -2.1149196655034745, V = -4.5691014376830479, S = -3.9833860032220842, N = -2.772588722;
    double mm[625], gg[25];
    emissions_kmer_setMatchProbsToDefaults(mm);
    emissions_kmer_setGapProbsToDefaults(gg);
    /* rows AA, AC, CA, NA of impl/emissionMatrix.c:22-47 */
    CHECK(mm[0] == M + M && mm[1] == M + V && mm[2] == M + S && mm[3] == M + V && mm[4] == M + N && mm[5] == V + M);
    CHECK(mm[25 + 0] == M + V && mm[25 + 1] == M + M && mm[25 + 3] == M + S && mm[25 + 9] == V + N);
    CHECK(mm[5 * 25 + 0] == V + M && mm[5 * 25 + 5] == M + M && mm[5 * 25 + 12] == V + S);
    CHECK(mm[20 * 25 + 0] == N + M && mm[20 * 25 + 2] == N + S && mm[24 * 25 + 24] == N + N);
    for (int a = 0; a < 25; a++)
        for (int b = 0; b < 25; b++) CHECK(mm[a * 25 + b] == mm[b * 25 + a]);
    for (int i = 0; i < 25; i++) CHECK(gg[i] == -3.2188758248682006);

    StateMachine *sMv = getSignalStateMachine3Vanilla(model);
    char *ref = "ATGACACATT";
    Sequence *kmers = sequence_construct(sequence_correctSeqLength(10, event), ref, sequence_getKmer2);
    void *k1 = kmers->get(kmers->elements, 1);
    const double *tab = sMv->EMISSION_MATCH_PROBS;
    const int64_t ki = emissions_discrete_getKmerIndex((char *) k1 + 1), kb = emissions_discrete_getKmerIndex(k1);
    const double mu = tab[1 + ki * MODEL_PARAMS], sd = tab[1 + ki * MODEL_PARAMS + 1], nmu = tab[1 + ki * MODEL_PARAMS + 2],
                 nsd = tab[1 + ki * MODEL_PARAMS + 3];
    double ev[3] = { 60.3, 0.9, 0.004 };
    const double a = (ev[0] - mu) / sd;
    CLOSE(emissions_signal_logGaussMatchProb(tab, k1, ev), log(0.3989422804014327) - log(sd) + (-0.5 * a * a), 1e-12);
    const double rho = tab[0], xu = (ev[0] - mu) / sd, yu = (ev[1] - nmu) / nsd;
    CLOSE(emissions_signal_getBivariateGaussPdfMatchProb(tab, k1, ev),
          -1.8378770664093453 - log(sd * nsd * sqrt(1 - rho * rho)) +
              (-1 / (2 * (1 - rho * rho))) * (xu * xu + yu * yu - 2 * rho * xu * yu), 1e-12);
    const double lambda = ev[2] / 0.00332005312085;
    CLOSE(emissions_signal_getDurationProb(ev, 2), 3 * 0.1397619423751586 + 2 * log(lambda) - 0.69314718056 - 2 * lambda, 1e-12);
    int64_t bin = (int64_t) (fabs(mu - tab[1 + kb * MODEL_PARAMS]) / 0.5);
    bin = bin >= 30 ? 29 : bin;
    CHECK(emissions_signal_getKmerSkipProb(sMv, k1) == sMv->EMISSION_GAP_X_PROBS[bin]);
    CHECK(emissions_signal_getKmerSkipProb(sMv, k1) == emissions_signal_getBetaOrAlphaSkipProb(sMv, k1, 0));
    /* scaleModelNoiseOnly = scaleModel without the level mean */
    StateMachine *sA = getStrawManStateMachine3(model), *sB = getStrawManStateMachine3(model);
    emissions_signal_scaleModel(sA, 1.03, 4.0, 1.1, 0.9, 1.2);
    emissions_signal_scaleModelNoiseOnly(sB, 1.03, 4.0, 1.1, 0.9, 1.2);
    for (int64_t k = 0; k < NUM_OF_KMERS; k += 97) {
        const double *x = sA->EMISSION_MATCH_PROBS + 1 + k * MODEL_PARAMS, *y = sB->EMISSION_MATCH_PROBS + 1 + k * MODEL_PARAMS;
        CHECK(x[1] == y[1] && x[2] == y[2] && x[3] == y[3] && x[4] == y[4]);
        CHECK(y[0] == tab[1 + k * MODEL_PARAMS] && x[0] == y[0] * 1.03 + 4.0);
    }
    stateMachine3_setTransitionsToNucleotideDefaults(sA);
    StateMachine3 *s3 = (StateMachine3 *) sA;
    CHECK(s3->TRANSITION_MATCH_CONTINUE == -0.030064059121770816 && s3->TRANSITION_GAP_SWITCH_TO_Y == -4.910694825551255);
    CLOSE(exp(s3->TRANSITION_MATCH_CONTINUE) + exp(s3->TRANSITION_GAP_OPEN_X) + exp(s3->TRANSITION_GAP_OPEN_Y), 1.0, 0.01);
    CLOSE(exp(s3->TRANSITION_MATCH_FROM_GAP_X) + exp(s3->TRANSITION_GAP_EXTEND_X) + exp(s3->TRANSITION_GAP_SWITCH_TO_Y), 1.0, 0.01);

    Diagonal d = diagonal_construct(7, -3, 5);
    char *txt = diagonal_getString(d);
    CHECK(strcmp(txt, "Diagonal, xay: 7 xmyL -3, xmyR: 5") == 0);
    free(txt);
    stList *pairs = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    stList_append(pairs, stIntTuple_construct2(5, 9));
    stList_append(pairs, stIntTuple_construct2(1, 2));
    stList_append(pairs, stIntTuple_construct2(4, 4));
    stList_sort(pairs, sortByXPlusYCoordinate);
    CHECK(stIntTuple_get(stList_get(pairs, 0), 0) == 1 && stIntTuple_get(stList_get(pairs, 1), 0) == 4 &&
          stIntTuple_get(stList_get(pairs, 2), 0) == 5);
    stList_destruct(pairs);
    stList *triples = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    stList_append(triples, stIntTuple_construct3(100, 5, 9));
    stList_append(triples, stIntTuple_construct3(900, 1, 2));
    stList_sort(triples, sortByXPlusYCoordinate2);
    CHECK(stIntTuple_get(stList_get(triples, 0), 0) == 900);
    stList_destruct(triples);

    /* the echelon decode on the strawMan toy matrices: state s of a cell stands for s k-mers -- gap X (1) gives one
     * pair, gap Y (2) two; counted here from the same cells */
    {
        Sequence *events7 = sequence_construct(7, toyEvents7, sequence_getEvent);
        char *ref13 = "ACGATACGGACAT";
        Sequence *kx = sequence_construct(sequence_correctSeqLength(13, event), ref13, sequence_getKmer);
        StateMachine *sM = getStrawManStateMachine3(model);
        const int64_t lX = kx->length, lY = events7->length;
        DpMatrix *F = dpMatrix_construct(lX + lY, 3), *B = dpMatrix_construct(lX + lY, 3);
        stList *none = stList_construct();
        Band *band = band_construct(none, lX, lY, 2);
        BandIterator *it = bandIterator_construct(band);
        Diagonal diags[64];
        for (int64_t i = 0; i <= lX + lY; i++) {
            Diagonal di = bandIterator_getNext(it);
            diags[i] = di;
            dpDiagonal_zeroValues(dpMatrix_createDiagonal(B, di));
            dpDiagonal_zeroValues(dpMatrix_createDiagonal(F, di));
        }
        dpDiagonal_initialiseValues(dpMatrix_getDiagonal(F, 0), sM, sM->startStateProb);
        dpDiagonal_initialiseValues(dpMatrix_getDiagonal(B, lX + lY), sM, sM->endStateProb);
        for (int64_t i = 1; i <= lX + lY; i++) diagonalCalculationForward(sM, i, F, kx, events7);
        for (int64_t i = lX + lY; i > 0; i--) diagonalCalculationBackward(sM, i, B, kx, events7);
        const double total = cell_dotProduct2(dpDiagonal_getCell(dpMatrix_getDiagonal(F, lX + lY), lX - lY), sM, sM->endStateProb);
        PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
        p->threshold = 0.05;
        stList *got = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
        void *extraArgs[1] = { got };
        int64_t want = 0;
        for (int64_t i = 1; i <= lX + lY; i++) {
            diagonalCalculationMultiPosteriorMatchProbs(sM, i, F, B, kx, events7, total, p, extraArgs);
            DpDiagonal *f = dpMatrix_getDiagonal(F, i), *b = dpMatrix_getDiagonal(B, i);
            for (int64_t xmy = diagonal_getMinXmy(diags[i]); xmy <= diagonal_getMaxXmy(diags[i]); xmy += 2) {
                if (diagonal_getXCoordinate(i, xmy) <= 0 || diagonal_getYCoordinate(i, xmy) <= 0) continue;
                for (int st = 1; st < 3; st++)
                    if (exp(dpDiagonal_getCell(f, xmy)[st] + dpDiagonal_getCell(b, xmy)[st] - total) >= p->threshold) want += st;
            }
        }
        CHECK(want > 0 && stList_length(got) == want);
        stList_destruct(got);
        pairwiseAlignmentBandingParameters_destruct(p);
        for (int64_t i = 0; i <= lX + lY; i++) {
            dpMatrix_deleteDiagonal(F, i);
            dpMatrix_deleteDiagonal(B, i);
        }
        dpMatrix_destruct(F);
        dpMatrix_destruct(B);
        bandIterator_destruct(it);
        band_destruct(band);
        stList_destruct(none);
        stateMachine_destruct(sM);
        sequence_sequenceDestroy(kx);
        sequence_sequenceDestroy(events7);
    }
    sequence_sequenceDestroy(kmers);
    stateMachine_destruct(sMv);
    stateMachine_destruct(sA);
    stateMachine_destruct(sB);
    printf("%s small_helpers\n", failures ? "FAILED" : "ok");
}

static int run_cpu(const char *goldenDir) {
    char *model = path_in(goldenDir, "template_median68pA.model"), *nhdpFile = path_in(goldenDir, "testTemplate.nhdp");
    /* the struct members of the reference's headers are there and filled */
    StateMachine *sM = getStrawManStateMachine3(model);
    StateMachine3 *sM3 = (StateMachine3 *) sM;
    CHECK(sM->type == threeState && sM->stateNumber == 3 && sM->matchState == match && sM->parameterSetSize == NUM_OF_KMERS);
    CHECK(sM->startStateProb && sM->endStateProb && sM->raggedStartStateProb && sM->raggedEndStateProb && sM->cellCalculate);
    CHECK(sM->cellCalculateUpdateExpectations == cell_signal_updateTransAndKmerSkipExpectations);
    CHECK(sM3->getXGapProbFcn == emissions_kmer_getGapProb && sM3->getMatchProbFcn == emissions_signal_strawManGetKmerEventMatchProb);
    CHECK(sM->startStateProb(sM, match) == 0 && sM->startStateProb(sM, shortGapX) == LOG_ZERO);
    CHECK(sM->raggedStartStateProb(sM, shortGapY) == 0 && sM->raggedStartStateProb(sM, match) == LOG_ZERO);
    CHECK(sM->endStateProb(sM, shortGapX) == sM3->TRANSITION_MATCH_FROM_GAP_X);
    CHECK(sM->raggedEndStateProb(sM, match) == (sM3->TRANSITION_GAP_OPEN_X + sM3->TRANSITION_GAP_OPEN_Y) / 2.0);
    printf("%s stateMachine_members\n", failures ? "FAILED" : "ok");
    test_small_helpers(model);

    char *ref10 = "ATGACACATT";
    Sequence *events5 = sequence_construct(5, toyEvents5, sequence_getEvent);
    Sequence *kmers = sequence_construct(sequence_correctSeqLength(10, event), ref10, sequence_getKmer);
    CHECK(kmers->length == 5);
    cell_round(sM, kmers->get(kmers->elements, 0), events5->get(events5->elements, 0), 0.00001, "strawMan_cell");
    sequence_sequenceDestroy(kmers);

    StateMachine *sMv = getSignalStateMachine3Vanilla(model);
    CHECK(sMv->type == vanilla && sMv->cellCalculateUpdateExpectations == cell_signal_updateBetaAndAlphaProb);
    CHECK(((StateMachine3Vanilla *) sMv)->getKmerSkipProb == emissions_signal_getBetaOrAlphaSkipProb);
    kmers = sequence_construct(sequence_correctSeqLength(10, event), ref10, sequence_getKmer2);
    cell_round(sMv, kmers->get(kmers->elements, 1), events5->get(events5->elements, 1), 0.00001, "vanilla_cell");
    sequence_sequenceDestroy(kmers);

    StateMachine *sM5 = stateMachine5_construct(fiveState, SYMBOL_NUMBER_NO_N, emissions_symbol_setEmissionsToDefaults,
                                                emissions_symbol_getGapProb, emissions_symbol_getGapProb,
                                                emissions_symbol_getMatchProb, cell_updateExpectations);
    CHECK(sM5->cellCalculateUpdateExpectations == cell_updateExpectations && ((StateMachine5 *) sM5)->getMatchProbFcn == emissions_symbol_getMatchProb);
    cell_round(sM5, "A", "A", 0.00001, "fiveState_cell");

    NanoporeHDP *nhdp = deserialize_nhdp(nhdpFile);
    StateMachine *sMh = getHdpStateMachine3(nhdp);
    CHECK(sMh->type == threeStateHdp && ((StateMachine3_HDP *) sMh)->hdpModel == nhdp);
    CHECK(((StateMachine3_HDP *) sMh)->getMatchProbFcn == get_nanopore_kmer_density);
    kmers = sequence_construct(sequence_correctSeqLength(10, event), ref10, sequence_getKmer3);
    double density = get_nanopore_kmer_density(nhdp, kmers->get(kmers->elements, 0), events5->get(events5->elements, 0));
    CHECK(density >= 0.0 && isfinite(density));
    printf("ok hdp_density %.17g\n", density);
    sequence_sequenceDestroy(kmers);
    sequence_sequenceDestroy(events5);

    test_dp_diagonal_and_matrix();

    /* test_diagonalDPCalculations pairwiseAlignerTest.c:278: AGCG against AGTTCG, 4 pairs at 0.2 */
    double total;
    Sequence *bX = sequence_construct(4, "AGCG", sequence_getBase), *bY = sequence_construct(6, "AGTTCG", sequence_getBase);
    stList *pairs = toy_dp(sM5, bX, bY, 0.2, &total, NULL, 2, 0);
    const int64_t want5[4][2] = { { 0, 0 }, { 1, 1 }, { 2, 4 }, { 3, 5 } };
    expect_pairs(pairs, want5, 4, "fiveState_diagonalDPCalculations");
    stList_destruct(pairs);
    sequence_sequenceDestroy(bX);
    sequence_sequenceDestroy(bY);

    /* test_strawMan_diagonalDPCalculations :580: 8 pairs at 0.2; test_vanilla_diagonalDPCalculations :795: 5 at 0.5 */
    char *ref13 = "ACGATACGGACAT";
    Sequence *events7 = sequence_construct(7, toyEvents7, sequence_getEvent);
    kmers = sequence_construct(sequence_correctSeqLength(13, event), ref13, sequence_getKmer);
    pairs = toy_dp(sM, kmers, events7, 0.2, &total, NULL, 2, 0);
    const int64_t want3[8][2] = { { 0, 0 }, { 1, 1 }, { 2, 2 }, { 3, 3 }, { 4, 3 }, { 5, 4 }, { 6, 5 }, { 7, 6 } };
    expect_pairs(pairs, want3, 8, "strawMan_diagonalDPCalculations");
    stList_destruct(pairs);
    sequence_sequenceDestroy(kmers);
    kmers = sequence_construct(sequence_correctSeqLength(13, event), ref13, sequence_getKmer2);
    pairs = toy_dp(sMv, kmers, events7, 0.5, &total, NULL, 2, 0);
    const int64_t wantV[5][2] = { { 2, 0 }, { 3, 3 }, { 5, 4 }, { 6, 5 }, { 7, 6 } };
    expect_pairs(pairs, wantV, 5, "vanilla_diagonalDPCalculations");
    stList_destruct(pairs);
    sequence_sequenceDestroy(kmers);
    sequence_sequenceDestroy(events7);

    /* plug-in constructors: a caller assembling the strawMan machine itself gets the same object */
    StateMachine *own = stateMachine3_construct(threeState, NUM_OF_KMERS, stateMachine3_setTransitionsToNanoporeDefaults,
                                                emissions_signal_initEmissionsToZero, emissions_kmer_getGapProb,
                                                emissions_signal_strawManGetKmerEventMatchProb,
                                                emissions_signal_strawManGetKmerEventMatchProb,
                                                cell_signal_updateTransAndKmerSkipExpectations);
    CHECK(own->cellCalculate == sM->cellCalculate && own->EMISSION_GAP_X_PROBS[17] == sM->EMISSION_GAP_X_PROBS[17]);
    CHECK(own->EMISSION_MATCH_PROBS[1] == 0.0 && ((StateMachine3 *) own)->TRANSITION_GAP_OPEN_X == sM3->TRANSITION_GAP_OPEN_X);
    stateMachine_destruct(own);
    own = stateMachine3Vanilla_construct(vanilla, NUM_OF_KMERS, emissions_signal_initEmissionsToZero,
                                         emissions_signal_getBetaOrAlphaSkipProb, emissions_signal_getEventMatchProbWithTwoDists,
                                         emissions_signal_getEventMatchProbWithTwoDists, cell_signal_updateBetaAndAlphaProb);
    CHECK(own->cellCalculate == sMv->cellCalculate && own->endStateProb(own, match) == sMv->endStateProb(sMv, match));
    stateMachine_destruct(own);
    own = stateMachine3Hdp_construct(threeStateHdp, NUM_OF_KMERS, stateMachine3_setTransitionsToNanoporeDefaults,
                                     emissions_signal_initEmissionsToZero, nhdp, emissions_kmer_getGapProb,
                                     get_nanopore_kmer_density, get_nanopore_kmer_density,
                                     cell_signal_updateTransAndKmerSkipExpectations2);
    CHECK(own->cellCalculate == sMh->cellCalculate && ((StateMachine3_HDP *) own)->hdpModel == nhdp);
    stateMachine_destruct(own);
    printf("%s plugin_constructors\n", failures ? "FAILED" : "ok");

    test_hmm_containers(goldenDir);

    stateMachine_destruct(sM);
    stateMachine_destruct(sMv);
    stateMachine_destruct(sM5);
    stateMachine_destruct(sMh);
    destroy_nanopore_hdp(nhdp);
    free(model);
    free(nhdpFile);
    return failures;
}

/* ---- estep: getSignalExpectations (vanillaAlign.c:318-359) ---------------------------------------------------- */
static void *read_file(const char *path, size_t *bytes) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    *bytes = (size_t) ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = malloc(*bytes + 1);
    if (fread(buf, 1, *bytes, f) != *bytes) exit(2);
    buf[*bytes] = 0;
    fclose(f);
    return buf;
}
static StateMachine *buildStateMachine(const char *modelFile, NanoporeReadAdjustmentParameters npp, StateMachineType type,
                                       Strand strand, NanoporeHDP *nHdp) { /* vanillaAlign.c:104-141 */
    if (type == vanilla) {
        StateMachine *sM = getSignalStateMachine3Vanilla(modelFile);
        emissions_signal_scaleModel(sM, npp.scale, npp.shift, npp.var, npp.scale_sd, npp.var_sd);
        stateMachine3Vanilla_setStrandTransitionsToDefaults(sM, strand);
        return sM;
    }
    if (type == threeState) {
        StateMachine *sM = getStrawManStateMachine3(modelFile);
        emissions_signal_scaleModel(sM, npp.scale, npp.shift, npp.var, npp.scale_sd, npp.var_sd);
        return sM;
    }
    return getHdpStateMachine3(nHdp);
}
static int run_estep(char **a) {
    const StateMachineType type = (StateMachineType) atoi(a[0]);
    size_t bytes;
    char *trainingTarget = read_file(a[2], &bytes);
    while (bytes && (trainingTarget[bytes - 1] == '\n' || trainingTarget[bytes - 1] == ' ')) trainingTarget[--bytes] = 0;
    double *events = read_file(a[3], &bytes);
    const int64_t nEvents = (int64_t) (bytes / sizeof(double)) / NB_EVENT_PARAMS;
    int64_t *eventMap = read_file(a[5], &bytes);
    NanoporeReadAdjustmentParameters npp = { atof(a[6]), atof(a[7]), atof(a[8]), atof(a[9]), atof(a[10]) };
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    p->threshold = atof(a[11]);
    p->diagonalExpansion = 40;
    p->minDiagsBetweenTraceBack = 150;
    p->splitMatrixBiggerThanThis = 100 * 100;
    stList *unmappedAnchors = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    FILE *f = fopen(a[4], "r");
    long long ax, ay;
    while (f && fscanf(f, "%lld %lld", &ax, &ay) == 2) stList_append(unmappedAnchors, stIntTuple_construct2(ax, ay));
    if (f) fclose(f);

    NanoporeHDP *nHdp = type == threeStateHdp ? deserialize_nhdp(a[1]) : NULL;
    Hmm *hmmExpectations = hmmContinuous_getEmptyHmm(type, 0.0, p->threshold);
    Sequence *eventSequence = sequence_construct2(nEvents, events, sequence_getEvent, sequence_sliceEventSequence2);

    /* from here on: the body of getSignalExpectations */
    StateMachine *sM = buildStateMachine(a[1], npp, type, template, nHdp);
    int64_t lX = sequence_correctSeqLength((int64_t) strlen(trainingTarget), event);
    stList *remapped = nanopore_remapAnchorPairsWithOffset(unmappedAnchors, eventMap, 0);
    stList *filteredRemappedAnchors = filterToRemoveOverlap(remapped);
    Sequence *target;
    if (type == vanilla) {
        target = sequence_construct2(lX, trainingTarget, sequence_getKmer2, sequence_sliceNucleotideSequence2);
        vanillaHmm_implantMatchModelsintoHmm(sM, hmmExpectations);
    } else if (type == threeStateHdp) {
        target = sequence_construct2(lX, trainingTarget, sequence_getKmer3, sequence_sliceNucleotideSequence2);
    } else {
        target = sequence_construct2(lX, trainingTarget, sequence_getKmer, sequence_sliceNucleotideSequence2);
    }
    getExpectationsUsingAnchors(sM, hmmExpectations, target, eventSequence, filteredRemappedAnchors, p,
                                diagonalCalculation_Expectations, 1, 1);

    printf("anchors %lld\n", (long long) stList_length(filteredRemappedAnchors));
    printf("likelihood %.17g\n", hmmExpectations->likelihood);
    if (type == vanilla) {
        printf("bins");
        for (int64_t i = 0; i < 60; i++) printf(" %.17g", hmmExpectations->getTransitionsExpFcn(hmmExpectations, i, 0));
        printf("\n");
    } else {
        printf("transitions");
        for (int64_t i = 0; i < 9; i++) printf(" %.17g", hmmExpectations->getTransitionsExpFcn(hmmExpectations, i / 3, i % 3));
        printf("\n");
    }
    if (type == threeState) {
        printf("kmergap");
        for (int64_t i = 0; i < NUM_OF_KMERS; i++) printf(" %.17g", hmmExpectations->getEmissionExpFcn(hmmExpectations, 0, i, 0));
        printf("\n");
    }
    if (type == threeStateHdp) {
        HdpHmm *h = (HdpHmm *) hmmExpectations;
        printf("assignments %lld\n", (long long) hmmContinuous_howManyAssignments(hmmExpectations));
        for (int64_t i = 0; i < h->numberOfAssignments; i++) {
            const char *k = stList_get(h->kmerAssignments, i);
            const double *e = stList_get(h->eventAssignments, i);
            /* the pointers sit inside the caller's own sequences, as in the reference */
            printf("assign %.6s %.17g %lld %lld\n", k, *e, (long long) (k - trainingTarget), (long long) ((e - events) / NB_EVENT_PARAMS));
        }
    }
    hmmContinuous_writeToFile(a[12], hmmExpectations, type);
    /* the M-step as trainModels' caller does it: normalise (not for the HDP), load back through the file */
    if (type != threeStateHdp) hmmContinuous_normalize(hmmExpectations, type);
    hmmContinuous_destruct(hmmExpectations, type);
    stateMachine_destruct(sM);
    sequence_sequenceDestroy(target);
    sequence_sequenceDestroy(eventSequence);
    stList_destruct(filteredRemappedAnchors);
    stList_destruct(remapped);
    stList_destruct(unmappedAnchors);
    pairwiseAlignmentBandingParameters_destruct(p);
    if (nHdp) destroy_nanopore_hdp(nHdp);
    return 0;
}

/* ---- gpu: the exported host internals against the aligner entry points on the toy alignments ----------------- */
static int same_pairs(stList *host, stList *entry) {
    /* (probability, x, y) for (probability, x, y); the entry point hands its list back in the order the reference's
     * getAlignedPairsUsingAnchors does, which pops the per-region lists and so reverses the emission order
     * (impl/pairwiseAligner.c:1447-1455) */
    const int64_t n = stList_length(host);
    int same = n == stList_length(entry);
    for (int64_t i = 0; same && i < n; i++)
        for (int k = 0; k < 3; k++)
            same &= stIntTuple_get(stList_get(host, i), k) == stIntTuple_get(stList_get(entry, n - 1 - i), k);
    if (!same) {
        for (int64_t i = 0; i < n; i++)
            fprintf(stderr, "host  %lld %lld %lld\n", (long long) stIntTuple_get(stList_get(host, i), 0),
                    (long long) stIntTuple_get(stList_get(host, i), 1), (long long) stIntTuple_get(stList_get(host, i), 2));
        for (int64_t i = 0; i < stList_length(entry); i++)
            fprintf(stderr, "entry %lld %lld %lld\n", (long long) stIntTuple_get(stList_get(entry, i), 0),
                    (long long) stIntTuple_get(stList_get(entry, i), 1), (long long) stIntTuple_get(stList_get(entry, i), 2));
    }
    return same;
}
static void host_against_gpu(StateMachine *sM, Sequence *SsX, Sequence *SsY, Hmm *hostHmm, Hmm *gpuHmm, int nValues,
                             double (*value)(Hmm *, int), const char *name) {
    double total;
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    p->diagonalExpansion = 40; /* wider than the toy matrices: the band is the whole matrix, as in toy_dp */
    stList *host = toy_dp(sM, SsX, SsY, p->threshold, &total, hostHmm, p->diagonalExpansion, 1);
    stList *anchorPairs = stList_construct();
    stList *gpu = getAlignedPairsUsingAnchors(sM, SsX, SsY, anchorPairs, p, diagonalCalculationPosteriorMatchProbs, 0, 0);
    CHECK(stList_length(gpu) > 0);
    CHECK(same_pairs(host, gpu));
    getExpectationsUsingAnchors(sM, gpuHmm, SsX, SsY, anchorPairs, p, diagonalCalculation_Expectations, 0, 0);
    /* host: likelihood += total once per diagonal (:853); the GPU path does the same sum */
    CLOSE(hostHmm->likelihood, gpuHmm->likelihood, 1e-9 * fabs(hostHmm->likelihood));
    for (int i = 0; i < nValues; i++) CLOSE(value(hostHmm, i), value(gpuHmm, i), 1e-9 * fabs(value(hostHmm, i)) + 1e-300);
    printf("%s %s %lld pairs total %.17g\n", failures ? "FAILED" : "ok", name, (long long) stList_length(gpu), total);
    stList_destruct(host);
    stList_destruct(gpu);
    stList_destruct(anchorPairs);
    pairwiseAlignmentBandingParameters_destruct(p);
}
static double transition_value(Hmm *h, int i) { return h->getTransitionsExpFcn(h, i / h->stateNumber, i % h->stateNumber); }
static double pair_hmm_value(Hmm *h, int i) { return i < 9 ? transition_value(h, i) : h->getEmissionExpFcn(h, 0, i - 9, 0); }
static double bin_value(Hmm *h, int i) { return h->getTransitionsExpFcn(h, i, 0); }
static double discrete_value(Hmm *h, int i) {
    return i < 25 ? transition_value(h, i) : h->getEmissionExpFcn(h, (i - 25) / 16, ((i - 25) % 16) / 4, (i - 25) % 4);
}
/* template || complement as vanillaAlign's two OpenMP sections run them (vanillaAlign.c:737-800): two threads, each
 * with a machine of its own, aligning at the same time; every result equals the one a lone caller gets */
typedef struct {
    const char *model;
    int vanillaMachine, rounds, mismatches;
    stList *expected;
} StrandJob;
static stList *align_toy(const char *model, int vanillaMachine) {
    char *ref13 = "ACGATACGGACAT";
    StateMachine *sM = vanillaMachine ? getSignalStateMachine3Vanilla(model) : getStrawManStateMachine3(model);
    Sequence *kmers = kmer_sequence(ref13, vanillaMachine ? sequence_getKmer2 : sequence_getKmer);
    Sequence *events7 = sequence_construct2(7, toyEvents7, sequence_getEvent, sequence_sliceEventSequence2);
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    stList *anchorPairs = stList_construct();
    stList *pairs = getAlignedPairsUsingAnchors(sM, kmers, events7, anchorPairs, p, diagonalCalculationPosteriorMatchProbs, 1, 1);
    stList_destruct(anchorPairs);
    pairwiseAlignmentBandingParameters_destruct(p);
    sequence_sequenceDestroy(kmers);
    sequence_sequenceDestroy(events7);
    stateMachine_destruct(sM);
    return pairs;
}
static int identical_lists(stList *a, stList *b) {
    if (stList_length(a) != stList_length(b)) return 0;
    for (int64_t i = 0; i < stList_length(a); i++)
        for (int k = 0; k < 3; k++)
            if (stIntTuple_get(stList_get(a, i), k) != stIntTuple_get(stList_get(b, i), k)) return 0;
    return 1;
}
static void *strand_thread(void *arg) {
    StrandJob *job = arg;
    for (int r = 0; r < job->rounds; r++) {
        stList *got = align_toy(job->model, job->vanillaMachine);
        if (!identical_lists(got, job->expected)) job->mismatches++;
        stList_destruct(got);
    }
    return NULL;
}
static void two_strands_at_once(const char *model) {
    StrandJob jobs[2] = { { model, 0, 25, 0, align_toy(model, 0) }, { model, 1, 25, 0, align_toy(model, 1) } };
    CHECK(stList_length(jobs[0].expected) > 0 && stList_length(jobs[1].expected) > 0);
    pthread_t th[2];
    for (int i = 0; i < 2; i++) pthread_create(&th[i], NULL, strand_thread, &jobs[i]);
    for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
    CHECK(jobs[0].mismatches == 0 && jobs[1].mismatches == 0);
    stList_destruct(jobs[0].expected);
    stList_destruct(jobs[1].expected);
    printf("%s two_threads_two_machines\n", failures ? "FAILED" : "ok");
}

static int run_gpu(const char *goldenDir) {
    char *model = path_in(goldenDir, "template_median68pA.model");
    char *ref13 = "ACGATACGGACAT";
    Sequence *events7 = sequence_construct2(7, toyEvents7, sequence_getEvent, sequence_sliceEventSequence2);

    StateMachine *sM = getStrawManStateMachine3(model);
    Sequence *kmers = kmer_sequence(ref13, sequence_getKmer);
    Hmm *a = hmmContinuous_getEmptyHmm(threeState, 0.0, 0.0), *b = hmmContinuous_getEmptyHmm(threeState, 0.0, 0.0);
    host_against_gpu(sM, kmers, events7, a, b, 9 + NUM_OF_KMERS, pair_hmm_value, "strawMan_host_vs_gpu");
    hmmContinuous_destruct(a, threeState);
    hmmContinuous_destruct(b, threeState);
    sequence_sequenceDestroy(kmers);
    stateMachine_destruct(sM);

    sM = getSignalStateMachine3Vanilla(model);
    kmers = kmer_sequence(ref13, sequence_getKmer2);
    a = hmmContinuous_getEmptyHmm(vanilla, 0.0, 0.0);
    b = hmmContinuous_getEmptyHmm(vanilla, 0.0, 0.0);
    vanillaHmm_implantMatchModelsintoHmm(sM, a);
    vanillaHmm_implantMatchModelsintoHmm(sM, b);
    host_against_gpu(sM, kmers, events7, a, b, 60, bin_value, "vanilla_host_vs_gpu");
    hmmContinuous_destruct(a, vanilla);
    hmmContinuous_destruct(b, vanilla);
    sequence_sequenceDestroy(kmers);
    stateMachine_destruct(sM);

    sM = stateMachine5_construct(fiveState, SYMBOL_NUMBER_NO_N, emissions_symbol_setEmissionsToDefaults,
                                 emissions_symbol_getGapProb, emissions_symbol_getGapProb, emissions_symbol_getMatchProb,
                                 cell_updateExpectations);
    Sequence *bX = sequence_construct2(4, "AGCG", sequence_getBase, sequence_sliceNucleotideSequence2);
    Sequence *bY = sequence_construct2(6, "AGTTCG", sequence_getBase, sequence_sliceNucleotideSequence2);
    a = hmmDiscrete_constructEmpty(0.0, 5, SYMBOL_NUMBER_NO_N, fiveState, hmmDiscrete_addToTransitionExpectation,
                                   hmmDiscrete_setTransitionExpectation, hmmDiscrete_getTransitionExpectation,
                                   hmmDiscrete_addToEmissionExpectation, hmmDiscrete_setEmissionExpectation,
                                   hmmDiscrete_getEmissionExpectation, emissions_discrete_getBaseIndex);
    b = hmmDiscrete_constructEmpty(0.0, 5, SYMBOL_NUMBER_NO_N, fiveState, hmmDiscrete_addToTransitionExpectation,
                                   hmmDiscrete_setTransitionExpectation, hmmDiscrete_getTransitionExpectation,
                                   hmmDiscrete_addToEmissionExpectation, hmmDiscrete_setEmissionExpectation,
                                   hmmDiscrete_getEmissionExpectation, emissions_discrete_getBaseIndex);
    host_against_gpu(sM, bX, bY, a, b, 25 + 5 * 16, discrete_value, "fiveState_host_vs_gpu");
    hmmDiscrete_destruct(a);
    hmmDiscrete_destruct(b);
    sequence_sequenceDestroy(bX);
    sequence_sequenceDestroy(bY);
    stateMachine_destruct(sM);
    sequence_sequenceDestroy(events7);
    two_strands_at_once(model);
    free(model);
    return failures;
}

int main(int argc, char **argv) {
    if (argc == 3 && !strcmp(argv[1], "cpu")) return run_cpu(argv[2]) ? 1 : 0;
    if (argc == 3 && !strcmp(argv[1], "gpu")) return run_gpu(argv[2]) ? 1 : 0;
    if (argc == 15 && !strcmp(argv[1], "estep")) return run_estep(argv + 2);
    fprintf(stderr, "usage: %s cpu|gpu <goldenDir> | estep <13 arguments, see the header comment>\n", argv[0]);
    return 2;
}
