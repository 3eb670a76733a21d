/*
 * cpecan_em.h -- the Baum-Welch loop of the strawMan signal machine as a native host loop, one process per GPU.
 *
 * What it replaces in the reference: the trainer's iteration (scripts/trainModels.py:244-330) -- a pool of vanillaAlign
 * processes writing one .expectations file each (vanillaAlign.c:680-716), the files summed in Python
 * (scripts/trainModels.py:126-135, scripts/nanoporeLib.py:991-1028), the sum normalised
 * (continuousPairHmm_normalize, impl/continuousHmm.c:174-191) and loaded into the next iteration's state machine
 * (continuousPairHmm_loadTransitionsAndKmerGapProbs, :206-232).  Here every rank keeps its reads, band tables, rings
 * and per-read scaled emission tables resident in HBM; an iteration rewrites the nine transitions and the 4096
 * k-mer gap probabilities in place, runs the E-step kernels, sums the per-read blocks on the device, combines the
 * ranks with ONE all-reduce of 4106 doubles over RCCL, and normalises on the host -- no files, no gather on a master.
 */
#ifndef CPECAN_EM_H_
#define CPECAN_EM_H_

#include "cpecan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t device;      /* HIP device of this process */
    int32_t rank, world; /* world 1: no communication */
    const char *id_file; /* world > 1: a path every rank can read; rank 0 writes RCCL's unique id there (the ranks of a
                            torch.distributed.run launch share a file system; no second rendezvous service is needed),
                            tagged with the rendezvous nonce below, and removes it once the communicator exists */
    /* this rank's reads in the C-ABI's batch layout (cpecan_hip_batch_create); items[i].model_id indexes match_tables */
    const cpecan_item *items;
    int64_t n_items;
    const char *x_chars;
    int64_t n_x;
    const double *events; /* [n_events][3] */
    int64_t n_events;
    const int64_t *anchors; /* [n_anchor_pairs][2] */
    int64_t n_anchor_pairs;
    const double *const *match_tables; /* n_models tables of CPECAN_MODEL_TABLE_LEN doubles, each scaled for its read
                                          (emissions_signal_scaleModel) */
    int32_t n_models;
    const double *gap_y_table; /* CPECAN_MODEL_TABLE_LEN doubles (the .model file's third line; not scaled by the reference) */
    cpecan_band_params params;
} cpecan_em_input;

/* `iterations` Baum-Welch iterations from (transitions[9] in the C-ABI's order, gap_x[4096]); both are overwritten
 * with the final model.  `pseudocount` is added to every expectation once per read (a rank adds pseudocount x its
 * n_items before the ranks are combined): the reference runs one vanillaAlign process per read, each starting its Hmm
 * from the pseudocount (vanillaAlign.c:668-669), and sums their files -- so the result does not depend on how the reads
 * are spread over ranks.  running_likelihood[i] receives the
 * summed log-likelihood of iteration i (under the model the iteration started from).  Identical on every rank. */
int cpecan_em_run(const cpecan_em_input *in, int32_t iterations, double pseudocount, double *transitions,
                  double *gap_x, double *running_likelihood);
const char *cpecan_em_last_error(void);

/* The id-file rendezvous of world > 1 (cpecan_em_run, cpecan_em_comm_create).  Every rank must derive the same nonce
 * for the same rendezvous and a different one from any earlier run that used the path: by default it is a hash of the
 * launcher's MASTER_ADDR, MASTER_PORT and TORCHELASTIC_RUN_ID (or CPECAN_EM_NONCE) and the number of rendezvous the
 * process has made; a caller that has a better one (a value agreed through its own store) sets it here, on every rank,
 * before the first rendezvous.  A rank that dies before ncclCommInitRank leaves the others waiting inside RCCL: that
 * part is RCCL's own and has not run at world > 1 on hardware. */
void cpecan_em_set_rendezvous_nonce(uint64_t nonce);
/* the file exchange on its own (what the two functions above do before ncclCommInitRank; tests drive it with two
 * processes and no GPU): rank 0 publishes `n` bytes (<= 128), the others receive them or fail after timeout_ms */
int cpecan_em_rendezvous_exchange(const char *id_file, int rank, int world, void *bytes, int64_t n, int32_t timeout_ms);
void cpecan_em_rendezvous_done(const char *id_file, int rank);

/* The ranks' communicator on its own, for the host library's training loop (cpecan_trainModels, include/cpecan_api.h:
 * any of the four machines): create once per process, hand cpecan_em_comm_reduce and the communicator to
 * cpecan_trainModels as its `reduce` / `reduceArg`.  The vector goes to the device, through ONE ncclAllReduce (sum)
 * and back; world = 1 needs no id file. */
typedef struct cpecan_em_comm cpecan_em_comm;
int cpecan_em_comm_create(int device, int rank, int world, const char *id_file, cpecan_em_comm **out);
void cpecan_em_comm_reduce(void *comm /* cpecan_em_comm* */, double *values, int64_t n);
int cpecan_em_comm_destroy(cpecan_em_comm *comm);

#ifdef __cplusplus
}
#endif
#endif
