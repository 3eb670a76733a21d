/*
 * vanillaAlign.c -- the signal-align driver of the reference (vanillaAlign.c:361-805) on top of libcpecan_host.so:
 * same command line, same inputs (a .npRead file, a reference sequence file, a guide alignment in exonerate CIGAR
 * format on stdin), same outputs (the 15-column TSV appended to --posteriors, the summary line on stdout, or the two
 * .expectations / .hmm files).  The template and the complement strand are aligned by two threads, as the
 * reference's two OpenMP sections do; each thread has a context of its own in the host library, so the two
 * alignments run on the GPU side by side.
 *
 * Not carried over: --buildHDP / --alignments (training and re-sampling HDPs is outside this path; the option is
 * refused with a message), the echelon machine (the reference marks it as not working, impl/stateMachine.c:1617), and
 * expectations under the fourState machine (hmmContinuous_getEmptyHmm has no container for it in the reference either).
 *
 * The guide alignment comes through sonLib's cigarRead in the reference.  sonLib is not part of this build, so
 * the line format is read here directly: "cigar: <query> <qStart> <qEnd> <+|-> <target> <tStart> <tEnd> <+|-> <score>
 * {<M|I|D> <length>}...", query = the 2D read (contig2/start2/end2 in the reference's struct), target = the
 * reference sequence (contig1/start1/end1); M advances both, D the target only, I the query only.
 */
#include <dirent.h>
#include <getopt.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cpecan_api.h"

static void die(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
static void die(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    exit(1);
}

/* ---- the guide alignment ----------------------------------------------------------------------------------- */
typedef struct {
    char op; /* 'M', 'I' (query only), 'D' (target only) */
    int64_t length;
} CigarOp;
typedef struct {
    char contig1[256], contig2[256]; /* 1 = target (reference), 2 = query (read) */
    int64_t start1, end1, strand1, start2, end2, strand2;
    double score;
    CigarOp *ops;
    int64_t nOps;
} GuideAlignment;

static GuideAlignment *cigar_read(FILE *f) {
    GuideAlignment *pA = calloc(1, sizeof *pA);
    char s1, s2;
    long long a, b, c, d;
    if (fscanf(f, " cigar: %255s %lld %lld %c %255s %lld %lld %c %lf", pA->contig2, &a, &b, &s2, pA->contig1, &c, &d, &s1,
               &pA->score) != 9)
        die("vanillaAlign - ERROR: no guide alignment (exonerate cigar line) on stdin");
    pA->start2 = a; pA->end2 = b; pA->strand2 = s2 == '+';
    pA->start1 = c; pA->end1 = d; pA->strand1 = s1 == '+';
    int64_t cap = 16;
    pA->ops = malloc(sizeof(CigarOp) * (size_t) cap);
    char op;
    long long len;
    while (fscanf(f, " %c %lld", &op, &len) == 2) {
        if (op != 'M' && op != 'I' && op != 'D') die("vanillaAlign - ERROR: cigar operation '%c'", op);
        if (pA->nOps == cap) pA->ops = realloc(pA->ops, sizeof(CigarOp) * (size_t) (cap *= 2));
        pA->ops[pA->nOps].op = op;
        pA->ops[pA->nOps++].length = len;
    }
    return pA;
}
/* checkPairwiseAlignment: the operations account for both intervals */
static void cigar_check(const GuideAlignment *pA) {
    int64_t l1 = 0, l2 = 0;
    for (int64_t i = 0; i < pA->nOps; i++) {
        if (pA->ops[i].op != 'I') l1 += pA->ops[i].length;
        if (pA->ops[i].op != 'D') l2 += pA->ops[i].length;
    }
    if (l1 != llabs(pA->end1 - pA->start1) || l2 != llabs(pA->end2 - pA->start2))
        die("vanillaAlign - ERROR: the cigar's operations cover %lld / %lld positions, its intervals %lld / %lld",
            (long long) l1, (long long) l2, (long long) llabs(pA->end1 - pA->start1), (long long) llabs(pA->end2 - pA->start2));
}

static char complement_of(char c) {
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;
    }
}
static char *reverse_complement(const char *s) {
    const size_t n = strlen(s);
    char *r = malloc(n + 1);
    for (size_t i = 0; i < n; i++) r[i] = complement_of(s[n - 1 - i]);
    r[n] = 0;
    return r;
}
static char *substring(const char *s, int64_t start, int64_t length) {
    if (start < 0 || length < 0 || (size_t) (start + length) > strlen(s))
        die("vanillaAlign - ERROR: the guide alignment reaches outside the reference sequence");
    char *r = malloc((size_t) length + 1);
    memcpy(r, s + start, (size_t) length);
    r[length] = 0;
    return r;
}
static char *replace_char(const char *s, char from, const char *to) { /* stString_replace(s, "C", to) */
    const size_t lt = strlen(to);
    size_t n = 0;
    for (const char *p = s; *p; p++) n += *p == from ? lt : 1;
    char *r = malloc(n + 1), *q = r;
    for (const char *p = s; *p; p++) {
        if (*p == from) { memcpy(q, to, lt); q += lt; }
        else *q++ = *p;
    }
    *q = 0;
    return r;
}
static char *first_line(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) die("vanillaAlign - ERROR: cannot open %s", path);
    size_t cap = 1 << 16, n = 0;
    char *buf = malloc(cap);
    int ch;
    while ((ch = fgetc(f)) != EOF && ch != '\n') {
        if (n + 2 > cap) buf = realloc(buf, cap *= 2);
        buf[n++] = (char) ch;
    }
    buf[n] = 0;
    fclose(f);
    return buf;
}

/* guideAlignmentToRebasedAnchorPairs (vanillaAlign.c:278-298): the target interval re-based to 0 (and flipped to the
 * forward strand), match columns to (target, query) pairs with `trim` columns cut at both ends of every match block
 * (convertPairwiseForwardStrandAlignmentToAnchorPairs, impl/pairwiseAligner.c:1039-1063), sorted, overlap-filtered */
static stList *guide_to_anchor_pairs(GuideAlignment *pA, PairwiseAlignmentParameters *p) {
    const bool flip = !pA->strand1;
    const int64_t shift = pA->strand1 ? pA->start1 : pA->end1;
    pA->start1 -= shift;
    pA->end1 -= shift;
    if (flip) {
        pA->strand1 = !pA->strand1;
        const int64_t t = pA->end1;
        pA->end1 = pA->start1;
        pA->start1 = t;
    }
    cigar_check(pA);
    if (!pA->strand1 || !pA->strand2) die("vanillaAlign - ERROR: the read side of the guide alignment must be '+'");
    stList *pairs = stList_construct3(0, (void (*)(void *)) stIntTuple_destruct);
    int64_t j = pA->start1, k = pA->start2;
    for (int64_t i = 0; i < pA->nOps; i++) {
        const CigarOp *op = &pA->ops[i];
        if (op->op == 'M')
            for (int64_t l = p->constraintDiagonalTrim; l < op->length - p->constraintDiagonalTrim; l++)
                stList_append(pairs, stIntTuple_construct2(j + l, k + l));
        if (op->op != 'I') j += op->length;
        if (op->op != 'D') k += op->length;
    }
    stList_sort(pairs, stIntTuple_cmpFn);
    stList *filtered = filterToRemoveOverlap(pairs);
    stList_destruct(pairs);
    return filtered;
}

static Sequence *event_sequence_from_guide(double *events, int64_t queryStart, int64_t queryEnd, int64_t *eventMap) {
    const int64_t startIdx = eventMap[queryStart], endIdx = eventMap[queryEnd];
    if (endIdx < startIdx) /* the reference would build a Sequence of negative length here (vanillaAlign.c:300-314) */
        die("vanillaAlign - ERROR: the event map runs against the read (event %lld at read position %lld, %lld at %lld)",
            (long long) startIdx, (long long) queryStart, (long long) endIdx, (long long) queryEnd);
    return sequence_construct2(endIdx - startIdx, events + startIdx * NB_EVENT_PARAMS, sequence_getEvent,
                               sequence_sliceEventSequence2);
}

static StateMachine *build_state_machine(const char *modelFile, NanoporeReadAdjustmentParameters npp,
                                         StateMachineType type, Strand strand, NanoporeHDP *nHdp) {
    if (type == vanilla) {
        StateMachine *sM = getSignalStateMachine3Vanilla(modelFile);
        emissions_signal_scaleModel(sM, npp.scale, npp.shift, npp.var, npp.scale_sd, npp.var_sd);
        stateMachine3Vanilla_setStrandTransitionsToDefaults(sM, strand);
        return sM;
    }
    if (type == threeState || type == fourState) { /* vanillaAlign.c:104-140 */
        StateMachine *sM = type == fourState ? getStateMachine4(modelFile) : getStrawManStateMachine3(modelFile);
        emissions_signal_scaleModel(sM, npp.scale, npp.shift, npp.var, npp.scale_sd, npp.var_sd);
        return sM;
    }
    if (!nHdp) die("vanillaAlign - ERROR: the strawMan-HDP model needs --templateHdp and --complementHdp");
    return getHdpStateMachine3(nHdp);
}
static stList *remapped_anchor_pairs(stList *unmapped, int64_t *eventMap, int64_t mapOffset) {
    stList *remapped = nanopore_remapAnchorPairsWithOffset(unmapped, eventMap, mapOffset);
    stList *filtered = filterToRemoveOverlap(remapped);
    stList_destruct(remapped);
    return filtered;
}
static void *(*target_getter(StateMachineType type))(void *, int64_t) {
    return type == vanilla ? sequence_getKmer2 : type == threeStateHdp ? sequence_getKmer3 : sequence_getKmer;
}

/* ---- one strand: what one OpenMP section of the reference does ----------------------------------------------- */
typedef struct {
    /* in */
    Strand strand;
    StateMachineType type;
    const char *modelFile, *hmmFile, *expectationsFile, *posteriorProbsFile, *readLabel;
    NanoporeHDP *nHdp;
    NanoporeReadAdjustmentParameters npp;
    Sequence *eventSequence;
    double *allEvents;
    int64_t *eventMap, mapOffset, eventShift, referenceShift;
    char *target, *tsvTarget, *contig;
    bool forward;
    PairwiseAlignmentParameters *p;
    stList *anchorPairs;
    /* out */
    stList *alignedPairs;
    double posteriorScore;
    StateMachine *sM; /* kept for the TSV rows (the scaled match table) */
} StrandJob;

static void *strand_expectations(void *arg) { /* getSignalExpectations + hmmContinuous_writeToFile (:318-359, :680-716) */
    StrandJob *j = arg;
    const char *name = j->strand == template ? "template" : "complement";
    fprintf(stderr, "vanillaAlign - getting expectations for %s\n", name);
    Hmm *hmm = hmmContinuous_getEmptyHmm(j->type, 0.0001, j->p->threshold);
    StateMachine *sM = build_state_machine(j->modelFile, j->npp, j->type, j->strand, j->nHdp);
    if (j->hmmFile) {
        fprintf(stderr, "vanillaAlign - loading HMM from file, %s\n", j->hmmFile);
        hmmContinuous_loadSignalHmm(j->hmmFile, sM, j->type);
    }
    const int64_t lX = sequence_correctSeqLength((int64_t) strlen(j->target), event);
    stList *anchors = remapped_anchor_pairs(j->anchorPairs, j->eventMap, j->mapOffset);
    Sequence *target = sequence_construct2(lX, j->target, target_getter(j->type), sequence_sliceNucleotideSequence2);
    if (j->type == vanilla) vanillaHmm_implantMatchModelsintoHmm(sM, hmm);
    getExpectationsUsingAnchors(sM, hmm, target, j->eventSequence, anchors, j->p, diagonalCalculation_Expectations, 1, 1);
    fprintf(stderr, "vanillaAlign - writing expectations to file: %s\n", j->expectationsFile);
    if (j->type == threeStateHdp)
        fprintf(stderr, "vanillaAlign - got %lld HDP assignments\n", (long long) hmmContinuous_howManyAssignments(hmm));
    hmmContinuous_writeToFile(j->expectationsFile, hmm, j->type);
    hmmContinuous_destruct(hmm, j->type);
    sequence_sequenceDestroy(target);
    stList_destruct(anchors);
    stateMachine_destruct(sM);
    return NULL;
}

static void *strand_alignment(void *arg) { /* performSignalAlignment + writePosteriorProbs (:179-255, :737-790) */
    StrandJob *j = arg;
    fprintf(stderr, "vanillaAlign - starting %s alignment\n", j->strand == template ? "template" : "complement");
    StateMachine *sM = build_state_machine(j->modelFile, j->npp, j->type, j->strand, j->nHdp);
    if (j->hmmFile) {
        fprintf(stderr, "loading HMM from file, %s\n", j->hmmFile);
        hmmContinuous_loadSignalHmm(j->hmmFile, sM, sM->type);
    }
    const int64_t lX = sequence_correctSeqLength((int64_t) strlen(j->target), event);
    fprintf(stderr, "vanillaAlign - doing banded alignment\n");
    stList *anchors = remapped_anchor_pairs(j->anchorPairs, j->eventMap, j->mapOffset);
    Sequence *sX = sequence_construct2(lX, j->target, target_getter(j->type), sequence_sliceNucleotideSequence2);
    j->alignedPairs = getAlignedPairsUsingAnchors(sM, sX, j->eventSequence, anchors, j->p,
                                                  diagonalCalculationPosteriorMatchProbs, 1, 1);
    double total = 0.0; /* scoreByPosteriorProbabilityIgnoringGaps */
    for (int64_t i = 0; i < stList_length(j->alignedPairs); i++) total += (double) stIntTuple_get(stList_get(j->alignedPairs, i), 0);
    j->posteriorScore = 100.0 * total / ((double) stList_length(j->alignedPairs) * PAIR_ALIGNMENT_PROB_1);
    stList_sort(j->alignedPairs, sortByXPlusYCoordinate2);
    sequence_sequenceDestroy(sX);
    stList_destruct(anchors);
    j->sM = sM;
    return NULL;
}

/* ---- a directory of reads: what scripts/signalAlign.py does with a pool of vanillaAlign processes, one GPU batch here.
 * Every <name>.npRead of the directory is aligned with the guide alignment in <name>.cigar (the exonerate cigar line the
 * pipeline would pipe into the single-read driver); template and complement strands of all reads go through ONE call of
 * getAlignedPairsUsingAnchorsBatch; posteriors are written to <outDir>/<name>.tsv, the summary lines to stdout. --------- */
typedef struct {
    char *name, *trimmedRefSeq, *rcTrimmedRefSeq;
    NanoporeRead *npRead;
    GuideAlignment *pA;
    stList *anchorPairs;
    Sequence *events[2];
    int64_t nAnchors;
} BatchRead;

static int name_order(const void *a, const void *b) { return strcmp(*(char *const *) a, *(char *const *) b); }

static int align_directory(const char *dir, const char *outDir, const char *referenceSequence, StateMachineType sMtype,
                           const char *modelFiles[2], const char *hmmFiles[2], NanoporeHDP *nHdps[2],
                           PairwiseAlignmentParameters *p, const char *substitute) {
    DIR *d = opendir(dir);
    if (!d) die("vanillaAlign - ERROR: cannot open the directory %s", dir);
    char **names = NULL;
    int64_t n = 0, cap = 0;
    for (struct dirent *e; (e = readdir(d)) != NULL;) {
        const size_t l = strlen(e->d_name);
        if (l <= 7 || strcmp(e->d_name + l - 7, ".npRead") != 0) continue;
        if (n == cap) names = realloc(names, sizeof(char *) * (size_t) (cap = cap ? 2 * cap : 64));
        names[n] = malloc(l + 1);
        memcpy(names[n], e->d_name, l - 7);
        names[n++][l - 7] = 0;
    }
    closedir(d);
    if (n == 0) die("vanillaAlign - ERROR: no .npRead file in %s", dir);
    qsort(names, (size_t) n, sizeof(char *), name_order);
    fprintf(stderr, "vanillaAlign - %lld reads in %s\n", (long long) n, dir);

    BatchRead *reads = calloc((size_t) n, sizeof(BatchRead));
    StateMachine **sMs = malloc(sizeof(StateMachine *) * 2 * (size_t) n);
    Sequence **sXs = malloc(sizeof(Sequence *) * 2 * (size_t) n), **sYs = malloc(sizeof(Sequence *) * 2 * (size_t) n);
    stList **anchors = malloc(sizeof(stList *) * 2 * (size_t) n);
    char **targets = malloc(sizeof(char *) * 2 * (size_t) n);
    for (int64_t i = 0; i < n; i++) {
        BatchRead *r = &reads[i];
        r->name = names[i];
        const size_t room = strlen(dir) + strlen(names[i]) + 16;
        char *path = malloc(room);
        snprintf(path, room, "%s/%s.npRead", dir, names[i]);
        r->npRead = nanopore_loadNanoporeReadFromFile(path);
        if (sMtype == threeStateHdp) nanopore_descaleNanoporeRead(r->npRead);
        snprintf(path, room, "%s/%s.cigar", dir, names[i]);
        FILE *f = fopen(path, "r");
        if (!f) die("vanillaAlign - ERROR: no guide alignment %s", path);
        r->pA = cigar_read(f);
        fclose(f);
        free(path);
        GuideAlignment *pA = r->pA;
        r->trimmedRefSeq = pA->strand1 ? substring(referenceSequence, pA->start1, pA->end1 - pA->start1)
                                       : substring(referenceSequence, pA->end1, pA->start1 - pA->end1);
        if (!pA->strand1) {
            char *rc = reverse_complement(r->trimmedRefSeq);
            free(r->trimmedRefSeq);
            r->trimmedRefSeq = rc;
        }
        r->rcTrimmedRefSeq = reverse_complement(r->trimmedRefSeq);
        r->events[0] = event_sequence_from_guide(r->npRead->templateEvents, pA->start2, pA->end2, r->npRead->templateEventMap);
        r->events[1] = event_sequence_from_guide(r->npRead->complementEvents, pA->start2, pA->end2, r->npRead->complementEventMap);
        const int64_t start1 = pA->start1, end1 = pA->end1;
        const bool forward = pA->strand1;
        r->anchorPairs = guide_to_anchor_pairs(pA, p); /* (re-bases the target interval in pA) */
        r->nAnchors = stList_length(r->anchorPairs);
        pA->start1 = start1; /* the TSV rows need the original offsets */
        pA->end1 = end1;
        pA->strand1 = forward;
        for (int st = 0; st < 2; st++) {
            const int64_t k = 2 * i + st;
            const NanoporeReadAdjustmentParameters npp = st == 0 ? r->npRead->templateParams : r->npRead->complementParams;
            sMs[k] = build_state_machine(modelFiles[st], npp, sMtype, st == 0 ? template : complement, nHdps[st]);
            if (hmmFiles[st]) hmmContinuous_loadSignalHmm(hmmFiles[st], sMs[k], sMs[k]->type);
            /* (as in the single-read driver, --substitute only applies when expectations are collected) */
            targets[k] = st == 0 ? r->trimmedRefSeq : r->rcTrimmedRefSeq;
            sXs[k] = sequence_construct2(sequence_correctSeqLength((int64_t) strlen(targets[k]), event), targets[k],
                                         target_getter(sMtype), sequence_sliceNucleotideSequence2);
            sYs[k] = r->events[st];
            anchors[k] = remapped_anchor_pairs(r->anchorPairs, st == 0 ? r->npRead->templateEventMap : r->npRead->complementEventMap,
                                               pA->start2);
        }
    }
    fprintf(stderr, "vanillaAlign - aligning %lld strands as one batch\n", (long long) (2 * n));
    stList **pairs = getAlignedPairsUsingAnchorsBatch(2 * n, sMs, sXs, sYs, anchors, p, 1, 1);
    for (int64_t i = 0; i < n; i++) {
        BatchRead *r = &reads[i];
        GuideAlignment *pA = r->pA;
        const size_t room = strlen(outDir) + strlen(r->name) + 16;
        char *out = malloc(room);
        snprintf(out, room, "%s/%s.tsv", outDir, r->name);
        remove(out); /* (writePosteriorProbs appends) */
        double score[2];
        for (int st = 0; st < 2; st++) {
            const int64_t k = 2 * i + st;
            double total = 0.0; /* scoreByPosteriorProbabilityIgnoringGaps */
            for (int64_t q = 0; q < stList_length(pairs[k]); q++) total += (double) stIntTuple_get(stList_get(pairs[k], q), 0);
            score[st] = 100.0 * total / ((double) stList_length(pairs[k]) * PAIR_ALIGNMENT_PROB_1);
            stList_sort(pairs[k], sortByXPlusYCoordinate2);
            const NanoporeReadAdjustmentParameters npp = st == 0 ? r->npRead->templateParams : r->npRead->complementParams;
            const int64_t *map = st == 0 ? r->npRead->templateEventMap : r->npRead->complementEventMap;
            writePosteriorProbs(out, r->name, sMs[k]->EMISSION_MATCH_PROBS, npp.scale, npp.shift,
                                st == 0 ? r->npRead->templateEvents : r->npRead->complementEvents,
                                st == 0 ? r->trimmedRefSeq : r->rcTrimmedRefSeq, pA->strand1, pA->contig1, map[pA->start2],
                                st == 0 ? pA->start1 : pA->end1, pairs[k], st == 0 ? template : complement);
        }
        fprintf(stdout, "%s %lld\t%lld(%f)\t%lld(%f)\n", r->name, (long long) r->nAnchors, (long long) stList_length(pairs[2 * i]),
                score[0], (long long) stList_length(pairs[2 * i + 1]), score[1]);
        free(out);
    }
    for (int64_t k = 0; k < 2 * n; k++) {
        stList_destruct(pairs[k]);
        stList_destruct(anchors[k]);
        sequence_sequenceDestroy(sXs[k]);
        stateMachine_destruct(sMs[k]);
    }
    (void) substitute;
    for (int64_t i = 0; i < n; i++) {
        sequence_sequenceDestroy(reads[i].events[0]);
        sequence_sequenceDestroy(reads[i].events[1]);
        stList_destruct(reads[i].anchorPairs);
        nanopore_nanoporeReadDestruct(reads[i].npRead);
        free(reads[i].pA->ops);
        free(reads[i].pA);
        free(reads[i].trimmedRefSeq);
        free(reads[i].rcTrimmedRefSeq);
        free(reads[i].name);
    }
    free(pairs); free(reads); free(sMs); free(sXs); free(sYs); free(anchors); free(targets); free(names);
    fprintf(stderr, "vanillaAlign - SUCCESS: finished alignment of %lld reads, exiting\n", (long long) n);
    return 0;
}

int main(int argc, char *argv[]) {
    StateMachineType sMtype = vanilla;
    const char *npReadDir = NULL, *outDir = NULL;
    int64_t diagExpansion = 50, constraintTrim = 14;
    double threshold = 0.01;
    const char *templateModelFile = "../../cPecan/models/template_median68pA.model";
    const char *complementModelFile = "../../cPecan/models/complement_median68pA_pop2.model";
    const char *readLabel = NULL, *npReadFile = NULL, *targetFile = NULL, *posteriorProbsFile = NULL;
    const char *templateHmmFile = NULL, *complementHmmFile = NULL, *templateExpectationsFile = NULL;
    const char *complementExpectationsFile = NULL, *templateHdp = NULL, *complementHdp = NULL, *substitute = NULL;
    static struct option long_options[] = {
        { "help", no_argument, 0, 'h' }, { "strawMan", no_argument, 0, 's' }, { "sm3Hdp", no_argument, 0, 'd' },
        { "fourState", no_argument, 0, 'f' }, { "echelon", no_argument, 0, 'e' }, { "buildHDP", no_argument, 0, 'U' },
        { "HdpType", required_argument, 0, 'p' }, { "substitute", required_argument, 0, 'M' },
        { "alignments", required_argument, 0, 'a' }, { "templateModel", required_argument, 0, 'T' },
        { "complementModel", required_argument, 0, 'C' }, { "readLabel", required_argument, 0, 'L' },
        { "npRead", required_argument, 0, 'q' }, { "reference", required_argument, 0, 'r' },
        { "posteriors", required_argument, 0, 'u' }, { "inTemplateHmm", required_argument, 0, 'y' },
        { "inComplementHmm", required_argument, 0, 'z' }, { "templateHdp", required_argument, 0, 'v' },
        { "complementHdp", required_argument, 0, 'w' }, { "templateExpectations", required_argument, 0, 't' },
        { "complementExpectations", required_argument, 0, 'c' }, { "diagonalExpansion", required_argument, 0, 'x' },
        { "threshold", required_argument, 0, 'D' }, { "constraintTrim", required_argument, 0, 'm' },
        { "npReadDir", required_argument, 0, 'B' }, { "outDir", required_argument, 0, 'O' }, { 0, 0, 0, 0 } };
    int key;
    while ((key = getopt_long(argc, argv, "hsdfeUp:M:a:T:C:L:q:r:u:y:z:v:w:t:c:x:D:m:B:O:", long_options, NULL)) != -1) {
        switch (key) {
        case 's': sMtype = threeState; break;
        case 'd': sMtype = threeStateHdp; break;
        case 'f': sMtype = fourState; break;
        case 'e': die("vanillaAlign - the echelon machine is not on the GPU path");
        case 'U': case 'a': case 'p':
            die("vanillaAlign - building and re-sampling HDPs is outside this path (run the reference's --buildHDP)");
        case 'M': substitute = optarg; break;
        case 'T': templateModelFile = optarg; break;
        case 'C': complementModelFile = optarg; break;
        case 'L': readLabel = optarg; break;
        case 'q': npReadFile = optarg; break;
        case 'r': targetFile = optarg; break;
        case 'u': posteriorProbsFile = optarg; break;
        case 't': templateExpectationsFile = optarg; break;
        case 'c': complementExpectationsFile = optarg; break;
        case 'y': templateHmmFile = optarg; break;
        case 'z': complementHmmFile = optarg; break;
        case 'v': templateHdp = optarg; break;
        case 'w': complementHdp = optarg; break;
        case 'x': diagExpansion = atoll(optarg); break;
        case 'D': threshold = atof(optarg); break;
        case 'm': constraintTrim = atoll(optarg); break;
        case 'B': npReadDir = optarg; break;
        case 'O': outDir = optarg; break;
        default:
            fprintf(stderr, "vanillaAlign binary, meant to be used through the signalAlign program.\n");
            return 1;
        }
    }
    if (npReadDir) { /* additive: a directory of reads as one GPU batch */
        if (!targetFile || !outDir) die("vanillaAlign - ERROR: --npReadDir needs --reference and --outDir");
        if (templateExpectationsFile || complementExpectationsFile)
            die("vanillaAlign - ERROR: --npReadDir aligns; expectations are collected read by read (or by cpecan_trainModels)");
        if ((templateHdp != NULL) != (complementHdp != NULL)) die("Need to have template and complement HDPs");
        const char *modelFiles[2] = { templateModelFile, complementModelFile }, *hmmFiles[2] = { templateHmmFile, complementHmmFile };
        NanoporeHDP *nHdps[2] = { templateHdp ? deserialize_nhdp(templateHdp) : NULL, complementHdp ? deserialize_nhdp(complementHdp) : NULL };
        PairwiseAlignmentParameters *bp = pairwiseAlignmentBandingParameters_construct();
        bp->threshold = threshold;
        bp->constraintDiagonalTrim = constraintTrim;
        bp->diagonalExpansion = diagExpansion;
        char *reference = first_line(targetFile);
        const int rc = align_directory(npReadDir, outDir, reference, sMtype, modelFiles, hmmFiles, nHdps, bp, substitute);
        free(reference);
        pairwiseAlignmentBandingParameters_destruct(bp);
        if (nHdps[0]) destroy_nanopore_hdp(nHdps[0]);
        if (nHdps[1]) destroy_nanopore_hdp(nHdps[1]);
        return rc;
    }
    if (!npReadFile || !targetFile) die("vanillaAlign - ERROR: --npRead and --reference are required");
    if (!readLabel) readLabel = npReadFile;
    fprintf(stderr, "vanillaAlign - using %s model\n",
            sMtype == threeState ? "strawMan" : sMtype == vanilla ? "vanilla" : sMtype == fourState ? "fourState" : "strawMan-HDP");
    if ((templateHdp != NULL) != (complementHdp != NULL)) die("Need to have template and complement HDPs");
    NanoporeHDP *nHdpT = templateHdp ? deserialize_nhdp(templateHdp) : NULL;
    NanoporeHDP *nHdpC = complementHdp ? deserialize_nhdp(complementHdp) : NULL;

    char *referenceSequence = first_line(targetFile);
    NanoporeRead *npRead = nanopore_loadNanoporeReadFromFile(npReadFile);
    if (sMtype == threeStateHdp) {
        fprintf(stderr, "vanillaAlign - descaling Nanopore Events\n");
        nanopore_descaleNanoporeRead(npRead);
    }
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    p->threshold = threshold;
    p->constraintDiagonalTrim = constraintTrim;
    p->diagonalExpansion = diagExpansion;

    GuideAlignment *pA = cigar_read(stdin);
    char *trimmedRefSeq = pA->strand1 ? substring(referenceSequence, pA->start1, pA->end1 - pA->start1)
                                      : substring(referenceSequence, pA->end1, pA->start1 - pA->end1);
    if (!pA->strand1) {
        char *rc = reverse_complement(trimmedRefSeq);
        free(trimmedRefSeq);
        trimmedRefSeq = rc;
    }
    char *rc_trimmedRefSeq = reverse_complement(trimmedRefSeq);
    char *templateTargetSeq = substitute ? replace_char(trimmedRefSeq, 'C', substitute) : trimmedRefSeq;
    char *complementTargetSeq = substitute ? replace_char(rc_trimmedRefSeq, 'C', substitute) : rc_trimmedRefSeq;
    Sequence *tEventSequence = event_sequence_from_guide(npRead->templateEvents, pA->start2, pA->end2, npRead->templateEventMap);
    Sequence *cEventSequence = event_sequence_from_guide(npRead->complementEvents, pA->start2, pA->end2, npRead->complementEventMap);
    const int64_t rShiftT = pA->start1, rShiftC = pA->end1;
    const bool forward = pA->strand1;
    stList *anchorPairs = guide_to_anchor_pairs(pA, p);

    const bool expectations = templateExpectationsFile && complementExpectationsFile;
    if (expectations && sMtype == fourState)
        die("vanillaAlign - the fourState machine has no expectations (the reference has no Hmm container for it)");
    StrandJob jobs[2] = {
        { template, sMtype, templateModelFile, templateHmmFile, templateExpectationsFile, posteriorProbsFile, readLabel, nHdpT,
          npRead->templateParams, tEventSequence, npRead->templateEvents, npRead->templateEventMap, pA->start2,
          npRead->templateEventMap[pA->start2], rShiftT, expectations ? templateTargetSeq : trimmedRefSeq, trimmedRefSeq,
          pA->contig1, forward, p, anchorPairs, NULL, 0.0, NULL },
        { complement, sMtype, complementModelFile, complementHmmFile, complementExpectationsFile, posteriorProbsFile, readLabel,
          nHdpC, npRead->complementParams, cEventSequence, npRead->complementEvents, npRead->complementEventMap, pA->start2,
          npRead->complementEventMap[pA->start2], rShiftC, expectations ? complementTargetSeq : rc_trimmedRefSeq,
          rc_trimmedRefSeq, pA->contig1, forward, p, anchorPairs, NULL, 0.0, NULL } };
    pthread_t th[2];
    for (int i = 0; i < 2; i++) pthread_create(&th[i], NULL, expectations ? strand_expectations : strand_alignment, &jobs[i]);
    for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
    if (!expectations) {
        /* the reference's two sections append to the same file in whatever order they finish; here the rows are
         * written after both are done, template first */
        for (int i = 0; i < 2 && posteriorProbsFile; i++)
            writePosteriorProbs((char *) posteriorProbsFile, (char *) readLabel, jobs[i].sM->EMISSION_MATCH_PROBS,
                                jobs[i].npp.scale, jobs[i].npp.shift, jobs[i].allEvents, jobs[i].tsvTarget, forward,
                                pA->contig1, jobs[i].eventShift, jobs[i].referenceShift, jobs[i].alignedPairs,
                                jobs[i].strand);
        stateMachine_destruct(jobs[0].sM);
        stateMachine_destruct(jobs[1].sM);
        fprintf(stdout, "%s %lld\t%lld(%f)\t", readLabel, (long long) stList_length(anchorPairs),
                (long long) stList_length(jobs[0].alignedPairs), jobs[0].posteriorScore);
        fprintf(stdout, "%lld(%f)\n", (long long) stList_length(jobs[1].alignedPairs), jobs[1].posteriorScore);
        stList_destruct(jobs[0].alignedPairs);
        stList_destruct(jobs[1].alignedPairs);
        fprintf(stderr, "vanillaAlign - SUCCESS: finished alignment of query %s, exiting\n", readLabel);
    }
    sequence_sequenceDestroy(tEventSequence);
    sequence_sequenceDestroy(cEventSequence);
    stList_destruct(anchorPairs);
    pairwiseAlignmentBandingParameters_destruct(p);
    nanopore_nanoporeReadDestruct(npRead);
    if (nHdpT) destroy_nanopore_hdp(nHdpT);
    if (nHdpC) destroy_nanopore_hdp(nHdpC);
    return 0;
}
