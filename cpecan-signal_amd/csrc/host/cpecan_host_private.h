/* cpecan_host_private.h -- shared by the two C files of libcpecan_host.so (not installed) */
#ifndef CPECAN_HOST_PRIVATE_H_
#define CPECAN_HOST_PRIVATE_H_

#include "cpecan_api.h"

void cpecan_die(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));

/* NanoporeHDP as far as density queries need it (deserialize_nhdp, cpecan_api.c) */
struct _nanopore_hdp {
    char alphabet[32];
    int64_t alphabetSize, kmerLength, numDps, gridLength, nRows;
    double *grid, *y, *slope; /* y, slope: nRows x gridLength */
    int32_t *kmerRow;         /* per k-mer id: row of the nearest observed ancestor (impl/hdp.c:2588-2590) */
};

/* fills the function-pointer members of a freshly built machine (cpecan_internals.c) */
void cpecan_sm3_set_functions(StateMachine3 *s);
void cpecan_sm3hdp_set_functions(StateMachine3_HDP *s);
void cpecan_sm3vanilla_set_functions(StateMachine3Vanilla *s);
void cpecan_sm5_set_functions(StateMachine5 *s);
/* 1 if cellCalculate and the emission plug-ins are the library's own, i.e. the model the device code implements */
int cpecan_sm_functions_known(StateMachine *sM);

#endif
