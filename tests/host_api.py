"""ctypes view of libcpecan_host.so (include/cpecan_api.h), for the tests.  Mirrors how a C caller of
the reference (vanillaAlign.c:179-255) uses the API."""
import ctypes as C
import os

import numpy as np

from cpecan_load import ROOT

LIB_PATH = os.environ.get("CPECAN_HOST_LIB") or os.path.join(ROOT, "cpecan-signal_amd", "libcpecan_host.so")
NUM_KMERS = 4096


class Params(C.Structure):
    _fields_ = [("threshold", C.c_double), ("minDiagsBetweenTraceBack", C.c_int64),
                ("traceBackDiagonals", C.c_int64), ("diagonalExpansion", C.c_int64),
                ("constraintDiagonalTrim", C.c_int64), ("anchorMatrixBiggerThanThis", C.c_int64),
                ("repeatMaskMatrixBiggerThanThis", C.c_int64), ("splitMatrixBiggerThanThis", C.c_int64),
                ("alignAmbiguityCharacters", C.c_bool), ("gapGamma", C.c_float)]


class StateMachine(C.Structure):
    _fields_ = [("type", C.c_int), ("stateNumber", C.c_int64), ("matchState", C.c_int64),
                ("parameterSetSize", C.c_int64), ("EMISSION_MATCH_PROBS", C.POINTER(C.c_double)),
                ("EMISSION_GAP_X_PROBS", C.POINTER(C.c_double)),
                ("EMISSION_GAP_Y_PROBS", C.POINTER(C.c_double))] + [
                    (n, C.c_void_p) for n in ("startStateProb", "endStateProb", "raggedEndStateProb",
                                              "raggedStartStateProb", "cellCalculate",
                                              "cellCalculateUpdateExpectations")]


class StateMachine3(C.Structure):
    _fields_ = [("model", StateMachine)] + [(n, C.c_double) for n in (
        "TRANSITION_MATCH_CONTINUE", "TRANSITION_MATCH_FROM_GAP_X", "TRANSITION_MATCH_FROM_GAP_Y",
        "TRANSITION_GAP_OPEN_X", "TRANSITION_GAP_OPEN_Y", "TRANSITION_GAP_EXTEND_X",
        "TRANSITION_GAP_EXTEND_Y", "TRANSITION_GAP_SWITCH_TO_X", "TRANSITION_GAP_SWITCH_TO_Y")]


class Diagonal(C.Structure):
    _fields_ = [("xay", C.c_int64), ("xmyL", C.c_int64), ("xmyR", C.c_int64)]


class Expectations(C.Structure):
    _fields_ = [("likelihood", C.c_double), ("transitions", C.c_double * 9),
                ("individualKmerGapProbs", C.c_double * NUM_KMERS)]


EXPORTS = [
    "stList_construct", "stList_construct3", "stList_destruct", "stList_length", "stList_get",
    "stList_append", "stIntTuple_construct2", "stIntTuple_construct3", "stIntTuple_get",
    "stIntTuple_length", "stIntTuple_destruct", "sequence_construct", "sequence_construct2",
    "sequence_sliceNucleotideSequence2", "sequence_sliceEventSequence2", "sequence_sequenceDestroy",
    "sequence_getKmer", "sequence_getEvent", "sequence_correctSeqLength",
    "pairwiseAlignmentBandingParameters_construct", "pairwiseAlignmentBandingParameters_destruct",
    "getStrawManStateMachine3", "stateMachine3_setTransitionsToNanoporeDefaults",
    "emissions_signal_scaleModel", "emissions_discrete_getKmerIndex", "stateMachine_destruct",
    "diagonalCalculationPosteriorMatchProbs", "getAlignedPairsUsingAnchors",
    "getAlignedPairsWithoutBanding", "getSplitPoints", "cpecan_getSignalExpectationsUsingAnchors",
    "cpecan_pairHmmExpectations_normalize", "cpecan_pairHmmExpectations_load",
    "getAlignedPairsUsingAnchorsBatch", "sequence_getBase", "sequence_sliceNucleotideSequence",
    "stateMachine5_construct", "emissions_symbol_setEmissionsToDefaults", "emissions_symbol_getGapProb",
    "emissions_symbol_getMatchProb", "cell_updateExpectations", "sequence_getKmer2",
    "getSignalStateMachine3Vanilla", "stateMachine3Vanilla_setStrandTransitionsToDefaults", "getStateMachine4", "stateMachine4_construct",
    "sequence_getKmer3", "deserialize_nhdp", "destroy_nanopore_hdp", "get_nanopore_hdp_alphabet_size",
    "get_nanopore_hdp_alphabet", "get_nanopore_kmer_density", "getHdpStateMachine3",
    "getPosteriorProbsWithBanding", "filterToRemoveOverlap", "diagonal_construct", "diagonal_getXay",
    "diagonal_getMinXmy", "diagonal_getMaxXmy", "diagonal_getWidth", "diagonal_getXCoordinate",
    "diagonal_getYCoordinate", "diagonal_equals", "band_construct", "band_destruct",
    "bandIterator_construct", "bandIterator_destruct", "bandIterator_clone", "bandIterator_getNext",
    "bandIterator_getPrevious", "logAdd", "nanopore_loadNanoporeReadFromFile", "nanopore_remapAnchorPairs",
    "nanopore_remapAnchorPairsWithOffset", "nanopore_descaleNanoporeRead", "nanopore_nanoporeReadDestruct",
    "cpecan_pairHmmExpectations_write", "cpecan_pairHmmExpectations_read", "hmmDiscrete_constructEmpty",
    "hmmDiscrete_addToTransitionExpectation", "hmmDiscrete_setTransitionExpectation",
    "hmmDiscrete_getTransitionExpectation", "hmmDiscrete_addToEmissionExpectation",
    "hmmDiscrete_setEmissionExpectation", "hmmDiscrete_getEmissionExpectation",
    "hmmDiscrete_randomizeTransitions", "hmmDiscrete_randomizeEmissions", "hmmDiscrete_randomize",
    "hmmDiscrete_normalize2", "hmmDiscrete_write", "hmmDiscrete_loadFromFile", "hmmDiscrete_destruct",
    "emissions_discrete_getBaseIndex", "stateMachineFunctions_construct", "getStateMachine5",
    "diagonalCalculation_Expectations", "getExpectationsUsingAnchors", "getExpectations",
    "getIndelProbabilities", "reweightAlignedPairs", "reweightAlignedPairs2", "sequence_padSequence",
    "cpecan_getVanillaExpectationsUsingAnchors", "cpecan_vanillaExpectations_normalize",
    "cpecan_vanillaExpectations_load", "cpecan_hdpExpectations_construct", "cpecan_hdpExpectations_destruct",
    "cpecan_getHdpExpectationsUsingAnchors", "cpecan_hdpExpectations_load", "cpecan_hdpExpectations_write", "writePosteriorProbs",
    "getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps", "getAlignedPairs",
    "cpecan_vanillaExpectations_write", "cpecan_vanillaExpectations_read", "cpecan_hdpExpectations_read",
]


class HdpExpectations(C.Structure):
    _fields_ = [("likelihood", C.c_double), ("transitions", C.c_double * 9), ("threshold", C.c_double),
                ("numberOfAssignments", C.c_int64), ("capacity", C.c_int64),
                ("eventAssignments", C.POINTER(C.c_double)), ("kmerAssignments", C.POINTER(C.c_char)),
                ("assignmentXY", C.POINTER(C.c_int64))]


class VanillaExpectations(C.Structure):
    _fields_ = [("likelihood", C.c_double), ("kmerSkipBins", C.c_double * 60)]


class Hmm(C.Structure):
    _fields_ = [("likelihood", C.c_double), ("type", C.c_int), ("stateNumber", C.c_int64),
                ("symbolSetSize", C.c_int64), ("matrixSize", C.c_int64)] + [
                    (n, C.c_void_p) for n in ("addToTransitionExpectationFcn", "setTransitionFcn",
                                              "getTransitionsExpFcn", "addToEmissionExpectationFcn",
                                              "setEmissionExpectationFcn", "getEmissionExpFcn",
                                              "getElementIndexFcn")]


class HmmDiscrete(C.Structure):
    _fields_ = [("baseHmm", Hmm), ("transitions", C.POINTER(C.c_double)), ("emissions", C.POINTER(C.c_double))]


class ContinuousPairHmm(C.Structure):
    _fields_ = [("baseHmm", Hmm), ("transitions", C.POINTER(C.c_double)),
                ("individualKmerGapProbs", C.POINTER(C.c_double))]


class StateMachine5(C.Structure):
    _fields_ = [("model", StateMachine)] + [(n, C.c_double) for n in (
        "MATCH_CONTINUE", "MATCH_FROM_SHORT_GAP_X", "MATCH_FROM_LONG_GAP_X", "GAP_SHORT_OPEN_X",
        "GAP_SHORT_EXTEND_X", "GAP_SHORT_SWITCH_TO_X", "GAP_LONG_OPEN_X", "GAP_LONG_EXTEND_X",
        "GAP_LONG_SWITCH_TO_X", "MATCH_FROM_SHORT_GAP_Y", "MATCH_FROM_LONG_GAP_Y", "GAP_SHORT_OPEN_Y",
        "GAP_SHORT_EXTEND_Y", "GAP_SHORT_SWITCH_TO_Y", "GAP_LONG_OPEN_Y", "GAP_LONG_EXTEND_Y",
        "GAP_LONG_SWITCH_TO_Y")]


def new_hmm_discrete(pseudocount, symbols=4, hmm_type=0):
    """hmmDiscrete_constructEmpty with the reference's own accessor functions"""
    L = lib()
    return L.hmmDiscrete_constructEmpty(
        pseudocount, 5, symbols, hmm_type, fn_ptr("hmmDiscrete_addToTransitionExpectation"),
        fn_ptr("hmmDiscrete_setTransitionExpectation"), fn_ptr("hmmDiscrete_getTransitionExpectation"),
        fn_ptr("hmmDiscrete_addToEmissionExpectation"), fn_ptr("hmmDiscrete_setEmissionExpectation"),
        fn_ptr("hmmDiscrete_getEmissionExpectation"), fn_ptr("emissions_discrete_getBaseIndex"))


class AdjustmentParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("scale", "shift", "var", "scale_sd", "var_sd")]


class NanoporeRead(C.Structure):
    _fields_ = [("readLength", C.c_int64), ("nbTemplateEvents", C.c_int64), ("nbComplementEvents", C.c_int64),
                ("templateParams", AdjustmentParams), ("complementParams", AdjustmentParams),
                ("twoDread", C.c_char_p), ("templateEventMap", C.POINTER(C.c_int64)),
                ("templateEvents", C.POINTER(C.c_double)), ("complementEventMap", C.POINTER(C.c_int64)),
                ("complementEvents", C.POINTER(C.c_double)), ("scaled", C.c_bool)]

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.stList_construct3.restype = vp
        L.stList_construct3.argtypes = [C.c_int64, vp]
        L.stList_destruct.argtypes = [vp]
        L.stList_length.restype = C.c_int64
        L.stList_length.argtypes = [vp]
        L.stList_get.restype = vp
        L.stList_get.argtypes = [vp, C.c_int64]
        L.stList_append.argtypes = [vp, vp]
        L.stIntTuple_construct2.restype = vp
        L.stIntTuple_construct2.argtypes = [C.c_int64, C.c_int64]
        L.stIntTuple_get.restype = C.c_int64
        L.stIntTuple_get.argtypes = [vp, C.c_int64]
        L.sequence_construct2.restype = vp
        L.sequence_construct2.argtypes = [C.c_int64, vp, vp, vp]
        L.sequence_sequenceDestroy.argtypes = [vp]
        L.sequence_correctSeqLength.restype = C.c_int64
        L.sequence_correctSeqLength.argtypes = [C.c_int64, C.c_int]
        L.pairwiseAlignmentBandingParameters_construct.restype = C.POINTER(Params)
        L.pairwiseAlignmentBandingParameters_destruct.argtypes = [C.POINTER(Params)]
        L.getStrawManStateMachine3.restype = C.POINTER(StateMachine3)
        L.getStrawManStateMachine3.argtypes = [C.c_char_p]
        L.emissions_signal_scaleModel.argtypes = [vp] + [C.c_double] * 5
        L.emissions_discrete_getKmerIndex.restype = C.c_int64
        L.emissions_discrete_getKmerIndex.argtypes = [C.c_char_p]
        L.stateMachine_destruct.argtypes = [vp]
        L.getSignalStateMachine3Vanilla.restype = vp
        L.getSignalStateMachine3Vanilla.argtypes = [C.c_char_p]
        L.getStateMachine4.restype = vp
        L.getStateMachine4.argtypes = [C.c_char_p]
        L.stateMachine3Vanilla_setStrandTransitionsToDefaults.argtypes = [vp, C.c_int]
        L.diagonal_construct.restype = Diagonal
        L.diagonal_construct.argtypes = [C.c_int64] * 3
        for f in ("getXay", "getMinXmy", "getMaxXmy", "getWidth"):
            getattr(L, "diagonal_" + f).restype = C.c_int64
            getattr(L, "diagonal_" + f).argtypes = [Diagonal]
        for f in ("getXCoordinate", "getYCoordinate"):
            getattr(L, "diagonal_" + f).restype = C.c_int64
            getattr(L, "diagonal_" + f).argtypes = [C.c_int64, C.c_int64]
        L.diagonal_equals.restype = C.c_int64
        L.diagonal_equals.argtypes = [Diagonal, Diagonal]
        L.band_construct.restype = vp
        L.band_construct.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64]
        L.band_destruct.argtypes = [vp]
        L.bandIterator_construct.restype = vp
        L.bandIterator_construct.argtypes = [vp]
        L.bandIterator_clone.restype = vp
        L.bandIterator_clone.argtypes = [vp]
        L.bandIterator_destruct.argtypes = [vp]
        L.bandIterator_getNext.restype = Diagonal
        L.bandIterator_getNext.argtypes = [vp]
        L.bandIterator_getPrevious.restype = Diagonal
        L.bandIterator_getPrevious.argtypes = [vp]
        L.logAdd.restype = C.c_double
        L.logAdd.argtypes = [C.c_double, C.c_double]
        L.nanopore_loadNanoporeReadFromFile.restype = C.POINTER(NanoporeRead)
        L.nanopore_loadNanoporeReadFromFile.argtypes = [C.c_char_p]
        L.nanopore_remapAnchorPairs.restype = vp
        L.nanopore_remapAnchorPairs.argtypes = [vp, C.POINTER(C.c_int64)]
        L.nanopore_remapAnchorPairsWithOffset.restype = vp
        L.nanopore_remapAnchorPairsWithOffset.argtypes = [vp, C.POINTER(C.c_int64), C.c_int64]
        L.nanopore_descaleNanoporeRead.argtypes = [C.POINTER(NanoporeRead)]
        L.nanopore_nanoporeReadDestruct.argtypes = [C.POINTER(NanoporeRead)]
        HP = C.POINTER(HmmDiscrete)
        L.hmmDiscrete_constructEmpty.restype = HP
        L.hmmDiscrete_constructEmpty.argtypes = [C.c_double, C.c_int64, C.c_int64, C.c_int] + [vp] * 7
        for name in ("hmmDiscrete_addToTransitionExpectation", "hmmDiscrete_setTransitionExpectation"):
            getattr(L, name).argtypes = [HP, C.c_int64, C.c_int64, C.c_double]
        L.hmmDiscrete_getTransitionExpectation.restype = C.c_double
        L.hmmDiscrete_getTransitionExpectation.argtypes = [HP, C.c_int64, C.c_int64]
        for name in ("hmmDiscrete_addToEmissionExpectation", "hmmDiscrete_setEmissionExpectation"):
            getattr(L, name).argtypes = [HP, C.c_int64, C.c_int64, C.c_int64, C.c_double]
        L.hmmDiscrete_getEmissionExpectation.restype = C.c_double
        L.hmmDiscrete_getEmissionExpectation.argtypes = [HP, C.c_int64, C.c_int64, C.c_int64]
        for name in ("hmmDiscrete_randomize", "hmmDiscrete_destruct"):
            getattr(L, name).argtypes = [HP]
        L.hmmDiscrete_normalize2.argtypes = [HP, C.c_bool]
        L.hmmDiscrete_loadFromFile.restype = HP
        L.hmmDiscrete_loadFromFile.argtypes = [C.c_char_p]
        L.cpecan_pairHmmExpectations_read.restype = C.POINTER(Expectations)
        L.cpecan_pairHmmExpectations_read.argtypes = [C.c_char_p]
        L.emissions_discrete_getBaseIndex.restype = C.c_int64
        L.emissions_discrete_getBaseIndex.argtypes = [C.c_char_p]
        L.stateMachineFunctions_construct.restype = vp
        L.stateMachineFunctions_construct.argtypes = [vp, vp, vp]
        L.getStateMachine5.restype = C.POINTER(StateMachine5)
        L.getStateMachine5.argtypes = [HP, vp]
        L.getExpectationsUsingAnchors.argtypes = [vp, vp, vp, vp, vp, C.POINTER(Params), vp, C.c_bool, C.c_bool]
        L.hmmContinuous_getEmptyHmm.restype = vp
        L.hmmContinuous_getEmptyHmm.argtypes = [C.c_int, C.c_double, C.c_double]
        L.hmmContinuous_destruct.argtypes = [vp, C.c_int]
        L.cpecan_getVanillaExpectationsUsingAnchors.argtypes = [vp, C.POINTER(VanillaExpectations), vp, vp, vp,
                                                         C.POINTER(Params), C.c_bool, C.c_bool]
        L.cpecan_vanillaExpectations_normalize.argtypes = [C.POINTER(VanillaExpectations)]
        L.cpecan_vanillaExpectations_load.argtypes = [vp, C.POINTER(VanillaExpectations)]
        L.cpecan_hdpExpectations_construct.restype = C.POINTER(HdpExpectations)
        L.cpecan_hdpExpectations_construct.argtypes = [C.c_double, C.c_double]
        L.cpecan_hdpExpectations_destruct.argtypes = [C.POINTER(HdpExpectations)]
        L.cpecan_getHdpExpectationsUsingAnchors.argtypes = [vp, C.POINTER(HdpExpectations), vp, vp, vp, C.POINTER(Params),
                                                     C.c_bool, C.c_bool]
        L.cpecan_hdpExpectations_load.argtypes = [vp, C.POINTER(HdpExpectations)]
        L.cpecan_hdpExpectations_write.argtypes = [C.POINTER(HdpExpectations), vp]
        L.writePosteriorProbs.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_double, C.c_double,
                                          C.POINTER(C.c_double), C.c_char_p, C.c_bool, C.c_char_p, C.c_int64,
                                          C.c_int64, vp, C.c_int]
        L.getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps.argtypes = [
            vp, vp, vp, vp, C.POINTER(Params), C.c_bool, C.c_bool, vp, vp, vp]
        L.getAlignedPairs.restype = vp
        L.getAlignedPairs.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.POINTER(Params), vp, vp, vp, C.c_bool,
                                      C.c_bool]
        L.cpecan_vanillaExpectations_write.argtypes = [C.POINTER(VanillaExpectations), vp, vp]
        L.cpecan_vanillaExpectations_read.restype = C.POINTER(VanillaExpectations)
        L.cpecan_vanillaExpectations_read.argtypes = [C.c_char_p, vp]
        L.cpecan_hdpExpectations_read.restype = C.POINTER(HdpExpectations)
        L.cpecan_hdpExpectations_read.argtypes = [C.c_char_p]
        L.getIndelProbabilities.restype = C.POINTER(C.c_int64)
        L.getIndelProbabilities.argtypes = [vp, C.c_int64, C.c_bool]
        L.reweightAlignedPairs2.restype = vp
        L.reweightAlignedPairs2.argtypes = [vp, C.c_int64, C.c_int64, C.c_double]
        L.filterToRemoveOverlap.restype = vp
        L.filterToRemoveOverlap.argtypes = [vp]
        L.getPosteriorProbsWithBanding.argtypes = [vp, vp, vp, vp, C.POINTER(Params), C.c_bool, C.c_bool, vp, vp]
        L.deserialize_nhdp.restype = vp
        L.deserialize_nhdp.argtypes = [C.c_char_p]
        L.destroy_nanopore_hdp.argtypes = [vp]
        L.get_nanopore_hdp_alphabet_size.restype = C.c_int64
        L.get_nanopore_hdp_alphabet_size.argtypes = [vp]
        L.getHdpStateMachine3.restype = vp
        L.getHdpStateMachine3.argtypes = [vp]
        L.stateMachine5_construct.restype = vp
        L.stateMachine5_construct.argtypes = [C.c_int, C.c_int64, vp, vp, vp, vp, vp]
        L.getAlignedPairsUsingAnchors.restype = vp
        L.getAlignedPairsUsingAnchors.argtypes = [vp, vp, vp, vp, C.POINTER(Params), vp, C.c_bool, C.c_bool]
        L.getAlignedPairsWithoutBanding.restype = vp
        L.getAlignedPairsWithoutBanding.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.POINTER(Params),
                                                    vp, vp, vp, C.c_bool, C.c_bool]
        L.getSplitPoints.restype = vp
        L.getSplitPoints.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_bool, C.c_bool]
        L.cpecan_getSignalExpectationsUsingAnchors.argtypes = [vp, C.POINTER(Expectations), vp, vp, vp,
                                                        C.POINTER(Params), C.c_bool, C.c_bool]
        L.cpecan_pairHmmExpectations_normalize.argtypes = [C.POINTER(Expectations)]
        L.cpecan_pairHmmExpectations_load.argtypes = [vp, C.POINTER(Expectations)]
        L.getAlignedPairsUsingAnchorsBatch.restype = C.POINTER(vp)
        L.getAlignedPairsUsingAnchorsBatch.argtypes = [C.c_int64, C.POINTER(vp), C.POINTER(vp),
                                                       C.POINTER(vp), C.POINTER(vp), C.POINTER(Params),
                                                       C.c_bool, C.c_bool]
        _LIB = L
    return _LIB


def fn_ptr(name):
    return C.cast(getattr(lib(), name), C.c_void_p)


def make_anchor_list(anchors):
    L = lib()
    lst = L.stList_construct3(0, fn_ptr("stIntTuple_destruct"))
    for x, y in anchors:
        L.stList_append(lst, L.stIntTuple_construct2(int(x), int(y)))
    return lst


def list_to_array(lst, width=3):
    L = lib()
    n = L.stList_length(lst)
    out = np.zeros((n, width), np.int64)
    for i in range(n):
        t = L.stList_get(lst, i)
        for j in range(width):
            out[i, j] = L.stIntTuple_get(t, j)
    return out


class Read:
    """keeps the C buffers of one read alive: k-mer Sequence over chars, event Sequence over doubles"""

    def __init__(self, x_chars, events):
        L = lib()
        self.xbuf = C.create_string_buffer(x_chars if isinstance(x_chars, bytes) else x_chars.encode())
        self.ev = np.ascontiguousarray(events, dtype=np.float64).reshape(-1)
        self.lX = L.sequence_correctSeqLength(len(self.xbuf.value), 2)
        self.lY = self.ev.size // 3
        self.sX = L.sequence_construct2(self.lX, C.cast(self.xbuf, C.c_void_p), fn_ptr("sequence_getKmer"),
                                        fn_ptr("sequence_sliceNucleotideSequence2"))
        self.sY = L.sequence_construct2(self.lY, self.ev.ctypes.data_as(C.c_void_p),
                                        fn_ptr("sequence_getEvent"), fn_ptr("sequence_sliceEventSequence2"))
