/*
 * cpecan_em.hip -- cpecan_em_run (include/cpecan_em.h): the Baum-Welch loop as a native host loop over the C-ABI,
 * with the ranks' expectation vectors combined by one RCCL all-reduce per iteration.  Host control code plus one
 * small reduction kernel; built into libcpecan_em.so (links libcpecan_hip.so and librccl).
 */
#include "cpecan_em.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define EM_HIP(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) { rc = fail(CPECAN_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)
#define EM_ABI(expr)                                                                          \
    do {                                                                                      \
        int r_ = (expr);                                                                      \
        if (r_ != CPECAN_OK) { rc = fail(r_, "%s: %s", #expr, cpecan_hip_last_error()); goto done; } \
    } while (0)
#define EM_NCCL(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t n_ = (expr);                                                                           \
        if (n_ != ncclSuccess) { rc = fail(CPECAN_EHIP, "%s: %s", #expr, ncclGetErrorString(n_)); goto done; } \
    } while (0)

/* out[i] = pseudocount (the caller's, times this rank's reads) + sum over the batch's models of their block's entry i (the likelihood, last entry, starts
 * from 0): the per-read expectation blocks of an E-step, added up where they are */
__global__ void cpecan_k_em_sum_blocks(const double *blocks, int nModels, int len, double pseudocount, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double s = i == len - 1 ? 0.0 : pseudocount;
    for (int m = 0; m < nModels; m++) s += blocks[(long long) m * len + i];
    out[i] = s;
}

/* continuousPairHmm_normalize (impl/continuousHmm.c:174-191) + continuousPairHmm_loadTransitionsAndKmerGapProbs
 * (:206-232) on one expectation vector [from * 3 + to | 4096 k-mer gaps | likelihood] */
void m_step(const double *e, double *transitions, double *gapX) {
    double t[9];
    for (int from = 0; from < 3; from++) {
        double total = 0.0;
        for (int to = 0; to < 3; to++) total += e[from * 3 + to];
        for (int to = 0; to < 3; to++) t[from * 3 + to] = e[from * 3 + to] / total;
    }
    double total = 0.0;
    for (int k = 0; k < CPECAN_NUM_KMERS; k++) total += e[9 + k];
    /* the C-ABI's order: MATCH_CONTINUE, MATCH_FROM_GAP_X, MATCH_FROM_GAP_Y, GAP_OPEN_X, GAP_OPEN_Y, GAP_EXTEND_X,
     * GAP_EXTEND_Y, GAP_SWITCH_TO_X, GAP_SWITCH_TO_Y; states match 0, gapX 1, gapY 2 */
    transitions[0] = log(t[0 * 3 + 0]);
    transitions[1] = log(t[1 * 3 + 0]);
    transitions[2] = log(t[2 * 3 + 0]);
    transitions[3] = log(t[0 * 3 + 1]);
    transitions[4] = log(t[0 * 3 + 2]);
    transitions[5] = log(1 - t[1 * 3 + 0]); /* sic: tied to MATCH_FROM_GAP_X (:217) */
    transitions[6] = log(t[2 * 3 + 2]);
    transitions[7] = log(t[2 * 3 + 1]);
    transitions[8] = -INFINITY; /* (:218) */
    for (int k = 0; k < CPECAN_NUM_KMERS; k++) gapX[k] = log(e[9 + k] / total);
}

/* Rank 0 hands RCCL's unique id to the other ranks through a file.  The file carries a nonce every rank derives the
 * same way (cpecan_em_set_rendezvous_nonce / CPECAN_EM_NONCE / the launcher's MASTER_ADDR:MASTER_PORT and run id, and
 * the number of rendezvous this process has made), so that a file left at the path by an earlier run, or by an
 * earlier rendezvous of this job, is not taken for this one's: rank 0 removes whatever is there before it publishes,
 * the other ranks accept a file only with their nonce and world size, and rank 0 removes it again once the communicator
 * exists (cpecan_em_comm_create / cpecan_em_run: ncclCommInitRank returns when every rank has joined). */
struct IdFile {
    char magic[8];
    uint64_t nonce;
    int32_t world, bytes;
    unsigned char id[128];
};
static_assert(sizeof(ncclUniqueId) <= 128, "id bytes");

std::mutex g_rdvLock;
uint64_t g_rdvBase = 0;
bool g_rdvBaseSet = false;
uint64_t g_rdvCalls = 0;

uint64_t fnv(uint64_t h, const char *s) {
    for (; s && *s; s++) h = (h ^ (unsigned char) *s) * 1099511628211ull;
    return h;
}

/* the nonce of this process's next rendezvous */
uint64_t next_nonce() {
    std::lock_guard<std::mutex> g(g_rdvLock);
    uint64_t base = g_rdvBase;
    if (!g_rdvBaseSet) {
        const char *e = getenv("CPECAN_EM_NONCE");
        if (e && *e) base = strtoull(e, nullptr, 0);
        else base = fnv(fnv(fnv(14695981039346656037ull, getenv("MASTER_ADDR")), getenv("MASTER_PORT")), getenv("TORCHELASTIC_RUN_ID"));
    }
    return base * 1000003ull + g_rdvCalls++;
}

int exchange_bytes(const char *path, int rank, int world, uint64_t nonce, void *id, size_t n, int timeoutMs) {
    if (n > sizeof(((IdFile *) nullptr)->id)) return fail(CPECAN_EINVAL, "id too long");
    if (rank == 0) {
        IdFile f;
        memset(&f, 0, sizeof f);
        memcpy(f.magic, "CPECANID", 8);
        f.nonce = nonce;
        f.world = world;
        f.bytes = (int32_t) n;
        memcpy(f.id, id, n);
        (void) unlink(path); /* whatever an earlier run left */
        const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long) getpid());
        FILE *o = fopen(tmp.c_str(), "wb");
        if (!o) return fail(CPECAN_EINVAL, "cannot write %s", tmp.c_str());
        const size_t w = fwrite(&f, 1, sizeof f, o);
        fclose(o);
        if (w != sizeof f || rename(tmp.c_str(), path) != 0) {
            (void) unlink(tmp.c_str());
            return fail(CPECAN_EINVAL, "cannot publish %s", path);
        }
        return CPECAN_OK;
    }
    for (int waited = 0; waited <= timeoutMs; waited += 50) {
        FILE *i = fopen(path, "rb");
        if (i) {
            IdFile f;
            const size_t r = fread(&f, 1, sizeof f, i);
            fclose(i);
            if (r == sizeof f && memcmp(f.magic, "CPECANID", 8) == 0 && f.nonce == nonce && f.world == world &&
                f.bytes == (int32_t) n) {
                memcpy(id, f.id, n);
                return CPECAN_OK;
            }
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    return fail(CPECAN_EINVAL, "rank %d: no RCCL id of this rendezvous appeared at %s", rank, path);
}

int exchange_id(const char *path, int rank, int world, ncclUniqueId *id) {
    return exchange_bytes(path, rank, world, next_nonce(), id, sizeof *id, 600000 /* ten minutes */);
}

void rendezvous_done(const char *path, int rank) {
    if (rank == 0 && path) (void) unlink(path);
}

} // namespace

extern "C" const char *cpecan_em_last_error(void) { return g_err.c_str(); }

extern "C" void cpecan_em_set_rendezvous_nonce(uint64_t nonce) {
    std::lock_guard<std::mutex> g(g_rdvLock);
    g_rdvBase = nonce;
    g_rdvBaseSet = true;
}

extern "C" int cpecan_em_rendezvous_exchange(const char *id_file, int rank, int world, void *bytes, int64_t n, int32_t timeout_ms) {
    if (!id_file || !bytes || n <= 0 || world < 2 || rank < 0 || rank >= world) return fail(CPECAN_EINVAL, "bad argument");
    return exchange_bytes(id_file, rank, world, next_nonce(), bytes, (size_t) n, timeout_ms);
}

extern "C" void cpecan_em_rendezvous_done(const char *id_file, int rank) { rendezvous_done(id_file, rank); }

extern "C" int cpecan_em_run(const cpecan_em_input *in, int32_t iterations, double pseudocount, double *transitions,
                             double *gapX, double *runningLikelihood) {
    if (!in || !transitions || !gapX || iterations < 0 || !in->items || in->n_items <= 0 || !in->match_tables ||
        in->n_models <= 0 || !in->gap_y_table || in->world < 1 || in->rank < 0 || in->rank >= in->world ||
        (in->world > 1 && !in->id_file))
        return fail(CPECAN_EINVAL, "bad argument");
    int rc = CPECAN_OK;
    cpecan_ctx *ctx = nullptr;
    cpecan_batch *batch = nullptr;
    ncclComm_t comm = nullptr;
    double *dSum = nullptr;
    void *stream = nullptr, *dBlocks = nullptr;
    int64_t nDoubles = 0;
    std::vector<cpecan_sm3_model> models((size_t) in->n_models);
    std::vector<int32_t> ids((size_t) in->n_models);
    std::vector<double> e(CPECAN_EXPECTATION_LEN);
    const int len = CPECAN_EXPECTATION_LEN;

    EM_HIP(hipSetDevice(in->device));
    if (in->world > 1) {
        ncclUniqueId id;
        if (in->rank == 0) EM_NCCL(ncclGetUniqueId(&id));
        rc = exchange_id(in->id_file, in->rank, in->world, &id);
        if (rc != CPECAN_OK) goto done;
        EM_NCCL(ncclCommInitRank(&comm, in->world, id, in->rank));
        rendezvous_done(in->id_file, in->rank);
    }
    EM_ABI(cpecan_hip_ctx_create(in->device, &ctx));
    EM_ABI(cpecan_hip_ctx_stream(ctx, &stream));
    for (int32_t m = 0; m < in->n_models; m++) {
        for (int k = 0; k < 9; k++) models[(size_t) m].transitions[k] = transitions[k];
        models[(size_t) m].match_probs = in->match_tables[m];
        models[(size_t) m].gap_x_probs = gapX;
        models[(size_t) m].gap_y_probs = in->gap_y_table;
    }
    EM_ABI(cpecan_hip_models_create(ctx, models.data(), in->n_models, 0, ids.data()));
    EM_ABI(cpecan_hip_batch_create(ctx, in->items, in->n_items, in->x_chars, in->n_x, in->events, in->n_events,
                                   in->anchors, in->n_anchor_pairs, &in->params, CPECAN_MODE_EXPECTATIONS,
                                   CPECAN_KERNEL_AUTO, 0, &batch));
    EM_ABI(cpecan_hip_batch_expectations_device_ptr(batch, &dBlocks, &nDoubles));
    if (nDoubles != (int64_t) in->n_models * len) {
        rc = fail(CPECAN_EINVAL, "expectation buffer holds %lld doubles, expected %lld", (long long) nDoubles,
                  (long long) in->n_models * len);
        goto done;
    }
    EM_HIP(hipMalloc((void **) &dSum, sizeof(double) * (size_t) len));

    for (int32_t it = 0; it < iterations; it++) {
        /* the model of this iteration, in place on the device; then the E-step of this rank's reads */
        EM_ABI(cpecan_hip_models_set_transitions(ctx, transitions, gapX));
        EM_ABI(cpecan_hip_batch_run(batch));
        EM_ABI(cpecan_hip_batch_sync(batch));
        hipLaunchKernelGGL(cpecan_k_em_sum_blocks, dim3((unsigned) (len + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                           (const double *) dBlocks, (int) in->n_models, len, pseudocount * (double) in->n_items, dSum);
        EM_HIP(hipGetLastError());
        if (comm) EM_NCCL(ncclAllReduce(dSum, dSum, (size_t) len, ncclDouble, ncclSum, comm, (hipStream_t) stream));
        EM_HIP(hipMemcpyAsync(e.data(), dSum, sizeof(double) * (size_t) len, hipMemcpyDeviceToHost, (hipStream_t) stream));
        EM_HIP(hipStreamSynchronize((hipStream_t) stream));
        if (runningLikelihood) runningLikelihood[it] = e[(size_t) len - 1];
        m_step(e.data(), transitions, gapX);
    }

done:
    if (dSum) (void) hipFree(dSum);
    if (batch) cpecan_hip_batch_destroy(batch);
    if (ctx) cpecan_hip_ctx_destroy(ctx);
    if (comm) (void) ncclCommDestroy(comm);
    return rc;
}

/* ---- the communicator on its own (for cpecan_trainModels of the host library) ---- */
struct cpecan_em_comm {
    int device = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *buf = nullptr;
    size_t cap = 0;
};

extern "C" int cpecan_em_comm_create(int device, int rank, int world, const char *idFile, cpecan_em_comm **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !idFile)) return fail(CPECAN_EINVAL, "bad argument");
    *out = nullptr;
    int rc = CPECAN_OK;
    cpecan_em_comm *c = new cpecan_em_comm();
    c->device = device;
    c->world = world;
    {
        ncclUniqueId id;
        EM_HIP(hipSetDevice(device));
        EM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        if (rank == 0) EM_NCCL(ncclGetUniqueId(&id));
        if (world > 1) {
            rc = exchange_id(idFile, rank, world, &id);
            if (rc != CPECAN_OK) goto done;
        }
        EM_NCCL(ncclCommInitRank(&c->comm, world, id, rank));
        if (world > 1) rendezvous_done(idFile, rank);
    }
done:
    if (rc != CPECAN_OK) {
        cpecan_em_comm_destroy(c);
        return rc;
    }
    *out = c;
    return CPECAN_OK;
}

extern "C" void cpecan_em_comm_reduce(void *arg, double *values, int64_t n) {
    cpecan_em_comm *c = (cpecan_em_comm *) arg;
    int rc = CPECAN_OK;
    if (!c || !values || n <= 0) return;
    EM_HIP(hipSetDevice(c->device));
    if (c->cap < (size_t) n) {
        if (c->buf) (void) hipFree(c->buf);
        c->buf = nullptr;
        c->cap = 0;
        EM_HIP(hipMalloc((void **) &c->buf, sizeof(double) * (size_t) n));
        c->cap = (size_t) n;
    }
    EM_HIP(hipMemcpyAsync(c->buf, values, sizeof(double) * (size_t) n, hipMemcpyHostToDevice, c->stream));
    EM_NCCL(ncclAllReduce(c->buf, c->buf, (size_t) n, ncclDouble, ncclSum, c->comm, c->stream));
    EM_HIP(hipMemcpyAsync(values, c->buf, sizeof(double) * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    EM_HIP(hipStreamSynchronize(c->stream));
done:
    if (rc != CPECAN_OK) { /* a training loop cannot continue with half a sum: say why and stop, as the host library does */
        fprintf(stderr, "cpecan_em_comm_reduce: %s\n", g_err.c_str());
        fflush(nullptr);
        _exit(1);
    }
}

extern "C" int cpecan_em_comm_destroy(cpecan_em_comm *c) {
    if (!c) return CPECAN_OK;
    (void) hipSetDevice(c->device);
    if (c->comm) (void) ncclCommDestroy(c->comm);
    if (c->buf) (void) hipFree(c->buf);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
    return CPECAN_OK;
}
