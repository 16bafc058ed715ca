// sharded.hpp - the stage loop over particle shards (one shard per GPU), included by smcmi.hip.
//
// Replaces the reference's only parallel mode - `@distributed` over particles with the whole cloud serialised to every worker
// each stage (src/smc_main.jl:169-170, 472-476) - by resident shards and a handful of tiny in-stream collectives:
//   per solver pass : all-reduce of 2 KC doubles (Σv, Σv² per candidate)      correction : all-reduce of (ΣW̃, ΣW̃²)
//   moments         : all-reduce of 1 + d + d(d+1)/2 doubles                  mutation   : all-reduce of Σ accept
//   selection       : (ESS < threshold only) all-gather of the weight column and of the shard clouds; every rank forms the
//                     same global cumulative sum and gathers the ancestors of ITS output slots.
// The kernels are the single-GPU ones: a pass prologue / k_post_correct / k_prepare_mutation / k_stage_begin that is handed
// the all-reduced totals as "partials of 1 block" takes exactly the same decision on every rank.
// Two communicators implement the collectives: RCCL (one process per GPU, xGMI; the library is dlopen'ed so that libsmcmi.so
// has no hard dependency on it) and an in-process group of handles (single-process multi-shard runs; used by the GPU tests
// to exercise this driver with 2-4 shards on one device).
#pragma once
#include <dlfcn.h>

// ---- minimal RCCL surface (rccl.h:40-43 and the five entry points used), resolved at run time
struct smcmi_nccl_uid { char internal[128]; };
typedef void *smcmi_nccl_comm;
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(smcmi_nccl_uid *) = nullptr;
    int (*CommInitRank)(smcmi_nccl_comm *, int, smcmi_nccl_uid, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, smcmi_nccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, smcmi_nccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, smcmi_nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, smcmi_nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(smcmi_nccl_comm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
enum { SMCMI_NCCL_DOUBLE = 8, SMCMI_NCCL_SUM = 0 };   // ncclFloat64, ncclSum

static int load_rccl() {
    if (g_rccl.lib) return 0;
    const char *cands[] = {getenv("SMCMI_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *c : cands) {
        if (!c || !*c) continue;
        g_rccl.lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return set_err(SMCMI_ERR_UNSUPPORTED, std::string("cannot load RCCL: ") + dlerror());
#define SMCMI_SYM(field, name) *(void **)(&g_rccl.field) = dlsym(g_rccl.lib, name)
    SMCMI_SYM(GetUniqueId, "ncclGetUniqueId"); SMCMI_SYM(CommInitRank, "ncclCommInitRank"); SMCMI_SYM(AllReduce, "ncclAllReduce");
    SMCMI_SYM(AllGather, "ncclAllGather"); SMCMI_SYM(Send, "ncclSend"); SMCMI_SYM(Recv, "ncclRecv");
    SMCMI_SYM(GroupStart, "ncclGroupStart"); SMCMI_SYM(GroupEnd, "ncclGroupEnd"); SMCMI_SYM(CommDestroy, "ncclCommDestroy"); SMCMI_SYM(GetErrorString, "ncclGetErrorString");
#undef SMCMI_SYM
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather)
        return set_err(SMCMI_ERR_UNSUPPORTED, "RCCL library lacks the required entry points");
    return 0;
}
#define NCCL_TRY(expr)                                                                                                    \
    do {                                                                                                                  \
        int r_ = (expr);                                                                                                  \
        if (r_ != 0) return set_err(SMCMI_ERR_HIP, std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "nccl error")); \
    } while (0)

static void smcmi_comm_release(smcmi_handle *h) {
    if (h->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(h->nccl);
    h->nccl = nullptr;
}

extern "C" int smcmi_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return set_err(SMCMI_ERR_ARG, "null argument");
    if (int rc = load_rccl()) return rc;
    smcmi_nccl_uid id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, 128);
    return 0;
}

extern "C" int smcmi_comm_init(smcmi_handle *h, int32_t rank, int32_t world, const uint8_t *id) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return set_err(SMCMI_ERR_ARG, "bad argument");
    if (h->cfg.n_local * world != h->cfg.n_parts || h->cfg.gid0 != rank * h->cfg.n_local)
        return set_err(SMCMI_ERR_ARG, "handle shard (n_local, gid0) does not match (rank, world): equal contiguous shards are required");
    if (int rc = load_rccl()) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device));
    smcmi_nccl_uid uid;
    memcpy(uid.internal, id, 128);
    smcmi_nccl_comm c = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&c, world, uid, rank));
    h->nccl = c; h->rank = rank; h->world = world;
    h->has_hostc = false;                             // (a handle connected to a host communicator before: RCCL carries the collectives now)
    h->mbox_tried = false; h->mbox_ok = false;        // the peer tables of the previous communicator's ranks are not this one's
    // self-test: every rank contributes (1, rank) -> (world, world (world - 1) / 2); a broken transport fails here, loudly,
    // instead of producing a wrong posterior later
    if (!h->d_comm || h->comm_cap < 2) return set_err(SMCMI_ERR_STATE, "communication scratch buffer missing");
    const double mine[2] = {1.0, (double)rank};
    double got[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(h->d_comm, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(g_rccl.AllReduce(h->d_comm, h->d_comm, 2, SMCMI_NCCL_DOUBLE, SMCMI_NCCL_SUM, h->nccl, h->stream));
    HIP_TRY(hipMemcpyAsync(got, h->d_comm, sizeof(got), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (got[0] != (double)world || got[1] != 0.5 * (double)world * (double)(world - 1))
        return set_err(SMCMI_ERR_HIP, "RCCL all-reduce self-test failed (got " + std::to_string(got[0]) + ", " + std::to_string(got[1]) + ")");
    return 0;
}

// ---- host-mediated communicator (include/smcmi.h smcmi_host_comm): the same collectives carried by caller-supplied functions on host
// buffers - for ranks that share a GPU or have no RCCL (the multi-process tests on one GPU, bring-up behind MPI / gloo / Distributed.jl).
// Every collective is: stream sync -> device-to-host -> callback -> host-to-device on the handle's stream -> sync, so kernels enqueued
// behind it see its result exactly as they see an in-stream RCCL collective's.
extern "C" int smcmi_comm_init_host(smcmi_handle *h, int32_t rank, int32_t world, const smcmi_host_comm *comm) {
    if (!h || !comm || !comm->allgather || !comm->barrier || world < 1 || rank < 0 || rank >= world) return set_err(SMCMI_ERR_ARG, "bad argument");
    if (h->cfg.n_local * world != h->cfg.n_parts || h->cfg.gid0 != rank * h->cfg.n_local)
        return set_err(SMCMI_ERR_ARG, "handle shard (n_local, gid0) does not match (rank, world): equal contiguous shards are required");
    if (h->nccl) smcmi_comm_release(h);
    h->has_hostc = false;
    h->mbox_tried = false; h->mbox_ok = false;
    // self-test, as smcmi_comm_init: every rank contributes (1, rank); the handle counts as connected only once it has passed
    const double mine[2] = {1.0, (double)rank};
    std::vector<double> all(2 * (size_t)world, 0.0);
    if (comm->allgather(mine, all.data(), 2, comm->user)) return set_err(SMCMI_ERR_CALLBACK, "host communicator: allgather failed");
    for (int r = 0; r < world; ++r)
        if (all[2 * r] != 1.0 || all[2 * r + 1] != (double)r) return set_err(SMCMI_ERR_CALLBACK, "host communicator: allgather self-test failed (rank order?)");
    h->hostc = *comm; h->has_hostc = true; h->rank = rank; h->world = world;
    return 0;
}
static int hostc_allgather(smcmi_handle *h, const double *dsend, double *drecv, size_t count) {
    std::vector<double> &sb = h->hc_send, &rb = h->hc_recv;
    sb.resize(count); rb.resize(count * (size_t)h->world);
    HIP_TRY(hipMemcpyAsync(sb.data(), dsend, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->hostc.allgather(sb.data(), rb.data(), (int64_t)count, h->hostc.user)) return set_err(SMCMI_ERR_CALLBACK, "host communicator: allgather failed");
    if (!drecv) return 0;                  // (the caller reads h->hc_recv on the host)
    HIP_TRY(hipMemcpyAsync(drecv, rb.data(), sizeof(double) * rb.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

constexpr int MAX_SHARDS = 64;     // slots for the per-shard energy maxima behind the all-reduced epilogue row
static inline int shard_rank(const smcmi_handle *h) { return (int)(h->cfg.gid0 / h->cfg.n_local); }

// ---- collectives over a group of lock-stepped local handles, optionally extended over ranks by RCCL
struct ShardGroup {
    std::vector<smcmi_handle *> hs;   // local shards (same process); RCCL mode: exactly one
    int world = 1;                    // total number of shards
    bool rccl = false;                // one local handle, the other shards in other processes ("remote")
    bool hostc = false;               // remote through the handle's host-mediated communicator instead of RCCL

    int sync_all() {
        for (auto *h : hs) { HIP_TRY(hipSetDevice(h->cfg.device)); HIP_TRY(hipStreamSynchronize(h->stream)); }
        return 0;
    }
    // in-place sum of `count` doubles at buf(h) over all shards, same result everywhere (fixed shard order)
    template <class F>
    int allreduce(F buf, int count) {
        if (rccl && hostc) {          // all-gather, then the sum in rank order (the same bits on every rank)
            smcmi_handle *h = hs[0];
            if (int rc = hostc_allgather(h, buf(h), nullptr, (size_t)count)) return rc;      // gathered on the host: h->hc_recv
            std::vector<double> tot(count, 0.0);
            for (int r = 0; r < world; ++r)
                for (int k = 0; k < count; ++k) tot[k] += h->hc_recv[(size_t)r * count + k];
            HIP_TRY(hipMemcpyAsync(buf(h), tot.data(), sizeof(double) * count, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            return 0;
        }
        if (rccl) {
            smcmi_handle *h = hs[0];
            NCCL_TRY(g_rccl.AllReduce(buf(h), buf(h), (size_t)count, SMCMI_NCCL_DOUBLE, SMCMI_NCCL_SUM, h->nccl, h->stream));
            return 0;
        }
        if (hs.size() == 1) return 0;
        if (int rc = sync_all()) return rc;
        std::vector<double> tot(count, 0.0), tmp(count);
        for (auto *h : hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            HIP_TRY(hipMemcpy(tmp.data(), buf(h), sizeof(double) * count, hipMemcpyDeviceToHost));
            for (int k = 0; k < count; ++k) tot[k] += tmp[k];
        }
        for (auto *h : hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            HIP_TRY(hipMemcpy(buf(h), tot.data(), sizeof(double) * count, hipMemcpyHostToDevice));
        }
        return 0;
    }
    // every handle's stream has reached this point before any goes on (RCCL: an in-stream all-reduce of one double)
    int barrier() {
        if (rccl && hostc) {
            smcmi_handle *h = hs[0];
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (h->hostc.barrier(h->hostc.user)) return set_err(SMCMI_ERR_CALLBACK, "host communicator: barrier failed");
            return 0;
        }
        if (rccl) {
            smcmi_handle *h = hs[0];
            NCCL_TRY(g_rccl.AllReduce(h->d_comm, h->d_comm, (size_t)1, SMCMI_NCCL_DOUBLE, SMCMI_NCCL_SUM, h->nccl, h->stream));
            return 0;
        }
        return sync_all();
    }
    // recv(h)[r * count .. (r+1) * count) = send(shard r)
    template <class FS, class FR>
    int allgather(FS send, FR recv, size_t count) {
        if (rccl && hostc) return hostc_allgather(hs[0], send(hs[0]), recv(hs[0]), count);
        if (rccl) {
            smcmi_handle *h = hs[0];
            NCCL_TRY(g_rccl.AllGather(send(h), recv(h), count, SMCMI_NCCL_DOUBLE, h->nccl, h->stream));
            return 0;
        }
        // (the copies go on the DESTINATION handle's stream, behind which its consumers are enqueued: the handles' streams are
        // non-blocking, a device-to-device hipMemcpy on the null stream is ordered with none of them and need not have finished when
        // it returns - consumers could read the table before the copy landed, and once in a few hundred runs did)
        if (int rc = sync_all()) return rc;
        for (auto *dst : hs) {
            HIP_TRY(hipSetDevice(dst->cfg.device));
            for (size_t r = 0; r < hs.size(); ++r)
                HIP_TRY(hipMemcpyAsync(recv(dst) + r * count, send(hs[r]), sizeof(double) * count, hipMemcpyDeviceToDevice, dst->stream));
        }
        return sync_all();                     // (and no source may go on before every copy out of it is done)
    }
    // Resample redistribution as an all-to-all-v (systematic resampling: the ancestors of a shard's output slots form one
    // contiguous global row range [lo[r], hi[r]]).  Every shard receives exactly the rows of that range, column by column, into
    // the [source shard][column][row] staging layout the all-gather would have produced, so k_resample_gather is unchanged.
    // ranges = (lo, hi) per shard, identical on every shard (all derive it from the same all-gathered cumulative weights).
    // with_cum: the rows' values of the handle's cum column (d_cum, local index) travel as one more column into d_cum_full (global index)
    int exchange_rows(const std::vector<long long> &ranges, bool with_cum = false) {
        const long long n = hs[0]->n;
        const int R = hs[0]->R;
        auto overlap = [&](int needer, int owner, long long &a, long long &b) {     // rows of `owner` that `needer` needs: [a, b)
            a = std::max(ranges[2 * needer], (long long)owner * n);
            b = std::min(ranges[2 * needer + 1] + 1, (long long)(owner + 1) * n);
            return b > a;
        };
        if (rccl && hostc) {
            // all-to-all-v on host buffers: per peer one [R][rows] block, packed from / unpacked into the column-major device layouts
            smcmi_handle *h = hs[0];
            const int me = h->rank;
            if (!h->hostc.alltoallv) return set_err(SMCMI_ERR_UNSUPPORTED, "host communicator lacks alltoallv (SMCMI_RESAMPLE_EXCHANGE=allgather avoids it)");
            const int RC = R + (with_cum ? 1 : 0);              // columns per row on the wire
            std::vector<int64_t> sc(world, 0), sd(world, 0), rcnt(world, 0), rd(world, 0);
            std::vector<long long> sa(world, 0), ra(world, 0);
            int64_t stot = 0, rtot = 0;
            for (int p = 0; p < world; ++p) {
                long long a, b;
                if (p != me && overlap(p, me, a, b)) { sc[p] = (b - a) * RC; sa[p] = a - (long long)me * n; }
                if (p != me && overlap(me, p, a, b)) { rcnt[p] = (b - a) * RC; ra[p] = a - (long long)p * n; }
                sd[p] = stot; stot += sc[p];
                rd[p] = rtot; rtot += rcnt[p];
            }
            std::vector<double> sbuf((size_t)std::max<int64_t>(stot, 1)), rbuf((size_t)std::max<int64_t>(rtot, 1));
            for (int p = 0; p < world; ++p)
                if (sc[p]) {
                    const long long len = sc[p] / RC;
                    HIP_TRY(hipMemcpy2DAsync(sbuf.data() + sd[p], sizeof(double) * len, h->cl.buf[0] + sa[p], sizeof(double) * n, sizeof(double) * (size_t)len, (size_t)R,
                                             hipMemcpyDeviceToHost, h->stream));
                    if (with_cum) HIP_TRY(hipMemcpyAsync(sbuf.data() + sd[p] + (size_t)R * len, h->d_cum + sa[p], sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, h->stream));
                }
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (h->hostc.alltoallv(sbuf.data(), sc.data(), sd.data(), rbuf.data(), rcnt.data(), rd.data(), h->hostc.user))
                return set_err(SMCMI_ERR_CALLBACK, "host communicator: alltoallv failed");
            for (int p = 0; p < world; ++p)
                if (rcnt[p]) {
                    const long long len = rcnt[p] / RC;
                    HIP_TRY(hipMemcpy2DAsync(h->d_full_cloud + (long long)p * R * n + ra[p], sizeof(double) * n, rbuf.data() + rd[p], sizeof(double) * len, sizeof(double) * (size_t)len,
                                             (size_t)R, hipMemcpyHostToDevice, h->stream));
                    if (with_cum) HIP_TRY(hipMemcpyAsync(h->d_cum_full + (long long)p * n + ra[p], rbuf.data() + rd[p] + (size_t)R * len, sizeof(double) * (size_t)len, hipMemcpyHostToDevice, h->stream));
                }
            long long a, b;
            if (overlap(me, me, a, b)) {
                HIP_TRY(hipMemcpy2DAsync(h->d_full_cloud + (long long)me * R * n + (a - (long long)me * n), sizeof(double) * n,
                                         h->cl.buf[0] + (a - (long long)me * n), sizeof(double) * n, sizeof(double) * (size_t)(b - a), (size_t)R,
                                         hipMemcpyDeviceToDevice, h->stream));
                if (with_cum) HIP_TRY(hipMemcpyAsync(h->d_cum_full + a, h->d_cum + (a - (long long)me * n), sizeof(double) * (size_t)(b - a), hipMemcpyDeviceToDevice, h->stream));
            }
            HIP_TRY(hipStreamSynchronize(h->stream));
            return 0;
        }
        if (rccl) {
            smcmi_handle *h = hs[0];
            const int me = h->rank;
            if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
                return set_err(SMCMI_ERR_UNSUPPORTED, "RCCL library lacks send/recv");
            NCCL_TRY(g_rccl.GroupStart());
            for (int p = 0; p < world; ++p) {
                long long a, b;
                if (overlap(p, me, a, b) && p != me) {                                 // my rows that p needs
                    for (int c = 0; c < R; ++c)
                        NCCL_TRY(g_rccl.Send(h->cl.buf[0] + (long long)c * n + (a - (long long)me * n), (size_t)(b - a), SMCMI_NCCL_DOUBLE, p, h->nccl, h->stream));
                    if (with_cum) NCCL_TRY(g_rccl.Send(h->d_cum + (a - (long long)me * n), (size_t)(b - a), SMCMI_NCCL_DOUBLE, p, h->nccl, h->stream));
                }
                if (overlap(me, p, a, b) && p != me) {                                 // p's rows that I need
                    for (int c = 0; c < R; ++c)
                        NCCL_TRY(g_rccl.Recv(h->d_full_cloud + ((long long)p * R + c) * n + (a - (long long)p * n), (size_t)(b - a), SMCMI_NCCL_DOUBLE, p, h->nccl, h->stream));
                    if (with_cum) NCCL_TRY(g_rccl.Recv(h->d_cum_full + a, (size_t)(b - a), SMCMI_NCCL_DOUBLE, p, h->nccl, h->stream));
                }
            }
            NCCL_TRY(g_rccl.GroupEnd());
            long long a, b;
            if (overlap(me, me, a, b)) {
                HIP_TRY(hipMemcpy2DAsync(h->d_full_cloud + (long long)me * R * n + (a - (long long)me * n), sizeof(double) * n,
                                         h->cl.buf[0] + (a - (long long)me * n), sizeof(double) * n, sizeof(double) * (size_t)(b - a), (size_t)R,
                                         hipMemcpyDeviceToDevice, h->stream));
                if (with_cum) HIP_TRY(hipMemcpyAsync(h->d_cum_full + a, h->d_cum + (a - (long long)me * n), sizeof(double) * (size_t)(b - a), hipMemcpyDeviceToDevice, h->stream));
            }
            return 0;
        }
        if (int rc = sync_all()) return rc;
        for (size_t d = 0; d < hs.size(); ++d)
            for (size_t o = 0; o < hs.size(); ++o) {
                long long a, b;
                if (!overlap((int)d, (int)o, a, b)) continue;
                HIP_TRY(hipSetDevice(hs[d]->cfg.device));
                HIP_TRY(hipMemcpy2DAsync(hs[d]->d_full_cloud + (long long)o * R * n + (a - (long long)o * n), sizeof(double) * n,
                                         hs[o]->cl.buf[0] + (a - (long long)o * n), sizeof(double) * n, sizeof(double) * (size_t)(b - a), (size_t)R,
                                         hipMemcpyDeviceToDevice, hs[d]->stream));
                if (with_cum) HIP_TRY(hipMemcpyAsync(hs[d]->d_cum_full + a, hs[o]->d_cum + (a - (long long)o * n), sizeof(double) * (size_t)(b - a), hipMemcpyDeviceToDevice, hs[d]->stream));
            }
        return sync_all();
    }
};

static int run2_guarded(ShardGroup &g, const smcmi_run_config *rc, smcmi_result *res);
static int ensure_shard_buffers(smcmi_handle *h) {
    const long long N = h->cfg.n_parts;
    if (!h->d_tot_ess) {
        if (dmalloc(&h->d_tot_ess, 2 * KC) || dmalloc(&h->d_tot_fin, 2) || dmalloc(&h->d_tot_mom, h->npairs + 2) || dmalloc(&h->d_tot_acc, ES + MAX_SHARDS))
            return SMCMI_ERR_HIP;
        HIP_TRY(hipMemsetAsync(h->d_tot_acc, 0, sizeof(double) * (ES + MAX_SHARDS), h->stream));   // (the handle's stream is non-blocking: keep its work on it)
    }
    if (!h->d_cum_full) {
        if (dmalloc(&h->d_cum_full, N)) return SMCMI_ERR_HIP;
        h->nb_full = (int)std::min<long long>(1024, std::max<long long>(1, (N + 511) / 512));
        if (dmalloc(&h->d_part_full, (size_t)h->nb_full * 2) || dmalloc(&h->d_off_full, h->nb_full)) return SMCMI_ERR_HIP;
    }
    if (!h->d_full_w) {
        if (dmalloc(&h->d_full_w, N) || dmalloc(&h->d_full_cloud, (size_t)N * h->R)) return SMCMI_ERR_HIP;
    }
    return 0;
}

// one stage, phase by phase over all local shards (the phases between collectives are independent per shard)
static int run_sharded_impl(ShardGroup &g, const smcmi_run_config *rc, smcmi_result *res) {
    smcmi_handle *h0 = g.hs[0];
    const int nf = h0->h_model.n_free;
    if (rc->n_blocks < 1 || rc->n_blocks > nf || ((nf + rc->n_blocks - 1) / rc->n_blocks) * (rc->n_blocks - 1) >= nf)
        return set_err(SMCMI_ERR_ARG, "n_blocks incompatible with the number of free parameters");
    if (rc->n_phi < 2 || rc->n_mh_steps < 1) return set_err(SMCMI_ERR_ARG, "bad n_phi / n_mh_steps");
    const bool adaptive = !rc->use_fixed_schedule;
    const int P_default = adaptive ? (rc->solver_passes >= 1 ? rc->solver_passes : SHARDED_SOLVER_PASSES) : 0;
    static const int no_pred = getenv("SMCMI_NO_PREDICTOR") ? atoi(getenv("SMCMI_NO_PREDICTOR")) : 0;   // development only
    // User likelihoods on the host (the reference's default use: smc(loglikelihood::Function, ...) with parallel = true, src/smc_main.jl:118,
    // 472-476): every shard scores the proposals of ITS particles through its registered callback (callback.hpp host_mutation: propose
    // kernel -> closure on the calling thread -> accept kernel), everything else - solver passes, correction, selection, moments,
    // proposal set-up, the collectives - is this driver's.  The callback mutation leaves neither energy sums nor energy maxima, so
    // such runs walk the schedule without a predictor, unshifted, one stage per host sync (the closure needs the host anyway).
    const bool host_mut = h0->cb[0] != nullptr, tempered_cb = h0->cb[1] != nullptr;
    for (auto *h : g.hs)
        if ((h->cb[0] != nullptr) != host_mut || (h->cb[1] != nullptr) != tempered_cb) return set_err(SMCMI_ERR_STATE, "every shard needs the same likelihood callbacks");
    const bool predict = adaptive && !no_pred && !host_mut;
    static const int sel_mode = getenv("SMCMI_NO_SELECT_PREDICT") ? atoi(getenv("SMCMI_NO_SELECT_PREDICT")) : 0;   // development only
    const bool predict_select = adaptive && can_fuse_post(h0) && sel_mode != 1 && !host_mut;
    std::vector<double> sched(rc->n_phi);
    for (int k = 0; k < rc->n_phi; ++k) sched[k] = pow((double)k / (double)(rc->n_phi - 1), rc->lambda);
    for (auto *h : g.hs) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        if (!adaptive && rc->n_phi > h->cfg.max_stages) return set_err(SMCMI_ERR_CAPACITY, "max_stages < n_phi");
        if (ensure_shard_buffers(h) || pull_state(h) || upload_sched(h, sched.data(), rc->n_phi)) return SMCMI_ERR_HIP;
        if (host_mut) { if (int e = ensure_callback_buffers(h)) return e; h->rng_ahead = false; h->cb_energy = false; h->cb_calls = 0; h->cb_evals = 0; }
        else if (int e = ensure_zbuf(h, rc->n_mh_steps, rc->n_blocks)) return e;
        DevState &s = h->h_st;
        RunParams rp{};
        rp.n_parts = h->cfg.n_parts; rp.n_blocks = rc->n_blocks; rp.n_mh_steps = rc->n_mh_steps; rp.n_phi = rc->n_phi;
        rp.resampling_method = rc->resampling_method; rp.use_fixed_schedule = rc->use_fixed_schedule;
        rp.threshold = rc->threshold_ratio * (double)h->cfg.n_parts;
        rp.alpha = rc->alpha; rp.target = rc->target; rp.tempering_target = rc->tempering_target;
        rp.pw = rc->tempered_update_prior_weight; rp.logp_old = rc->log_prob_old_data;
        rp.max_stages = h->cfg.max_stages; rp.store_history = h->cfg.store_history;
        rp.stall_on_exhaust = 1;
        rp.phi_rtol = rc->phi_rtol > 0.0 ? rc->phi_rtol : (rc->phi_rtol < 0.0 ? 0.0 : DEFAULT_PHI_RTOL);
        rp.stop_stage = rc->stop_after_stage > 0 ? rc->stop_after_stage : 0;
        if (rc->continue_run) {
            // continue_intermediate (smc_main.jl:334-335,355-361): every shard keeps the loop scalars, records and history it holds
            // (a paused run, or smcmi_set_loop_state / _set_stage_records / _set_history with the same scalars on every shard)
            if (s.stage < 1 || s.stage >= h->cfg.max_stages) return set_err(SMCMI_ERR_STATE, "no loop state to continue from");
            if (s.phi_n >= 1.0) return set_err(SMCMI_ERR_STATE, "the run to continue has already reached phi = 1");
            if (s.stage != h0->h_st.stage) return set_err(SMCMI_ERR_STATE, "shards hold different loop states");
            s.rp = rp; s.done = 0; s.err = 0; s.skip_fold = 1; s.do_resample = 0;
            if (push_state(h)) return SMCMI_ERR_HIP;
            continue;
        }
        memset(&s, 0, sizeof(DevState));
        s.e_seen = __builtin_nan("");
        s.rp = rp;
        s.stage = 1; s.j = 2; s.c = rc->c; s.accept = rc->target;
        s.ess_prev = rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts;
        if (push_state(h)) return SMCMI_ERR_HIP;
        const double v0[4] = {0.0, rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts, rc->c, rc->target};
        HIP_TRY(hipMemcpy(h->rec.phi, &v0[0], sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->rec.ess, &v0[1], sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->rec.c, &v0[2], sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->rec.accept, &v0[3], sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemsetAsync(h->rec.resampled, 0, sizeof(int) * h->cfg.max_stages, h->stream));
        if (h->cfg.store_history) {
            HIP_TRY(hipMemsetAsync(h->d_hist_w, 0, sizeof(double) * h->n, h->stream));
            HIP_TRY(hipMemcpyAsync(h->d_hist_W, h->cl.buf[0] + (long long)(h->R - 1) * h->n, sizeof(double) * h->n, hipMemcpyDeviceToDevice, h->stream));
        }
    }
    // stage 1's energy shift: largest energy of the initial cloud over all shards (slots after the ES row: a sum all-reduce
    // in which every shard fills only its own slot is a gather)
    if (g.world > MAX_SHARDS) return set_err(SMCMI_ERR_ARG, "too many shards");
    if (!host_mut) {
    for (auto *h : g.hs) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        const int nb0 = mut_blocks(h);
        k_energy_max<<<nb0, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_emax_part);
        k_reduce_partials<<<1, TB, 0, h->stream>>>(nullptr, 0, 0, nullptr, h->d_emax_part, nb0, h->d_tot_acc + ES, shard_rank(h), g.world);
    }
    if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_acc + ES; }, g.world)) return rc2;
    } else {                                  // stage 2's begin folds "the last acceptance rate": none yet
        for (auto *h : g.hs) { HIP_TRY(hipSetDevice(h->cfg.device)); HIP_TRY(hipMemsetAsync(h->d_tot_acc, 0, sizeof(double) * (ES + MAX_SHARDS), h->stream)); }
    }
    // One stage = the single-GPU launch sequence with the collectives in-stream.  Nothing in it needs the host: on an adaptive
    // schedule resampling is predictable (smcmi_run), so the selection path - all-gather of weights and shard clouds, global
    // scan, gather - is enqueued exactly where a resample is expected (its kernels gate themselves on the device's decision)
    // and elsewhere k_moments_reg takes the post-correction decision itself; a wrong expectation or a solver that runs out
    // of passes stalls the run (done = 3 / 2) identically on every rank and is resumed at the next host sync.  Fixed schedules
    // cannot be predicted: they keep one flag read per stage.
    // mode 0: full stage with selection path; 1: no selection expected; 2: tail only (from k_post_correct on; selection path)
    // spec: predict -> correct -> verify (no certificate pass, no all-reduce for it: 2 collectives per stage);
    // skip_begin: resume of such a stage through the certificate path
    // use_graph == 2: HIP events around the first local shard's mutation launches (the `roofline` figure of a multi-GPU bench line)
    const bool profile = rc->use_graph == 2;
    std::vector<hipEvent_t> mut_evs;
    auto enqueue = [&](int p0, int P, int mode, bool spec = false, bool skip_begin = false) -> int {
        if (spec) P = 0;
        const int fin_slot = P == 0 ? 0 : (P & 1);
        // no selection expected + register kernels: the correction pass gathers the moments too, so (ΣW̃, ΣW̃², pair sums) travel
        // in ONE all-reduce (3 collectives per stage instead of 4); k_prepare_mutation decides, the mutation kernel normalises
        const bool cm = mode == 1 && can_fuse_cm(h0);
        const int npf = h0->npairs + 2;
        for (auto *h : g.hs) { h->fused_cm = cm; h->spec_stage = spec && cm; }
        if (mode == 2) {
            // resume of a stalled stage: its correction left the per-block (ΣW̃, ΣW̃²) partials - total them first
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_part_fin, h->nb_e, 2, h->d_tot_fin);
            }
            if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_fin; }, 2)) return rc2;
        }
        if (mode != 2) {
            if (p0 == 0 && !skip_begin)
                for (auto *h : g.hs) {
                    HIP_TRY(hipSetDevice(h->cfg.device));
                    h->run_adaptive = predict;
                    k_stage_begin<<<1, BT, 0, h->stream>>>(h->d_st, h->d_sched, h->d_tot_acc + EACC, 1, h->rec, predict ? h->d_tot_acc : nullptr, nullptr,
                                                   h->spec_stage ? 1 : 0, host_mut ? nullptr : h->d_tot_acc + ES, g.world);
                }
            for (int p = p0; p < P; ++p) {
                for (auto *h : g.hs) {
                    HIP_TRY(hipSetDevice(h->cfg.device));
                    k_pass<KC, false><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, h->d_tot_ess, h->d_part_ess[p & 1], 1, p, nullptr, 0);
                    k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_part_ess[p & 1], h->nb_e, 2 * KC, h->d_tot_ess);
                }
                if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_ess; }, 2 * KC)) return rc2;
            }
            if (cm) {
                for (auto *h : g.hs) {
                    HIP_TRY(hipSetDevice(h->cfg.device));
                    launch_correct_moments(h, P, h->d_tot_ess, 1);
                    k_moments_reduce<<<(npf + 63) / 64, 1024, 0, h->stream>>>(h->d_st, h->d_part_cm, h->nb_e, npf, h->d_tot_mom, 0);
                }
                if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_mom; }, npf)) return rc2;
            } else {
                for (auto *h : g.hs) {
                    HIP_TRY(hipSetDevice(h->cfg.device));
                    k_pass<1, true><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, h->d_tot_ess, h->d_part_fin, 1, P, h->d_hist_w, h->n);
                    k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_part_fin, h->nb_e, 2, h->d_tot_fin);
                }
                if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_fin; }, 2)) return rc2;
            }
        }
        if (cm) {
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                launch_prepare_in_run(h, h->d_tot_mom, 1, 3, fin_slot);
                hipEvent_t pe0 = nullptr, pe1 = nullptr;
                if (profile && h == h0) { hipEventCreate(&pe0); hipEventCreate(&pe1); mut_evs.push_back(pe0); mut_evs.push_back(pe1); hipEventRecord(pe0, h->stream); }
                const int nbl = launch_mutate(h, rc->n_blocks, 0, rc->alpha);
                if (pe1) hipEventRecord(pe1, h->stream);
                if (predict) k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_esum_part, nbl, ES, h->d_tot_acc, h->d_emax_part, nbl, h->d_tot_acc + ES, shard_rank(h), g.world);
                else k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_acc_part, nbl, 1, h->d_tot_acc + EACC, h->d_emax_part, nbl, h->d_tot_acc + ES, shard_rank(h), g.world);
            }
            if (predict) { if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_acc; }, ES + g.world)) return rc2; }
            else if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_acc + EACC; }, 1 + g.world)) return rc2;
            return 0;
        }
        if (mode == 1) {
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                // the moments kernel decides from the all-reduced (ΣW̃, ΣW̃²) handed over as the partials of one block
                switch (h->d) {
#define SMCMI_FUSED_CASE(D) case D: k_moments_reg<D><<<h->nb_mr, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_part_mom, h->d_hist_W, h->n, 0, h->d_tot_fin, 1, fin_slot, h->rec); break;
                SMCMI_FUSED_CASE(1) SMCMI_FUSED_CASE(2) SMCMI_FUSED_CASE(3) SMCMI_FUSED_CASE(4) SMCMI_FUSED_CASE(5) SMCMI_FUSED_CASE(6)
                SMCMI_FUSED_CASE(7) SMCMI_FUSED_CASE(8) SMCMI_FUSED_CASE(9) SMCMI_FUSED_CASE(10) SMCMI_FUSED_CASE(11) SMCMI_FUSED_CASE(12)
#undef SMCMI_FUSED_CASE
                default: return set_err(SMCMI_ERR_STATE, "fused post-correction needs d <= 12");
                }
                k_moments_reduce<<<(h->npairs + 63) / 64, 1024, 0, h->stream>>>(h->d_st, h->d_part_mom, h->nb_mr, h->npairs, h->d_tot_mom, 0);
            }
        } else {
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                k_post_correct<<<1, TB, 0, h->stream>>>(h->d_st, h->d_tot_fin, 1, nullptr, h->rec, fin_slot);
            }
            int rs = 1;
            if (!adaptive || !predict_select) {
                // the one host decision per stage of the unpredicted path (identical on all ranks)
                int fl[2] = {0, 0};
                HIP_TRY(hipSetDevice(h0->cfg.device));
                HIP_TRY(hipMemcpyAsync(&fl[0], &h0->d_st->do_resample, sizeof(int), hipMemcpyDeviceToHost, h0->stream));
                HIP_TRY(hipMemcpyAsync(&fl[1], &h0->d_st->done, sizeof(int), hipMemcpyDeviceToHost, h0->stream));
                HIP_TRY(hipStreamSynchronize(h0->stream));
                rs = fl[0] && !fl[1];
            }
            // resample redistribution: all-to-all-v of exactly the rows each shard's slots descend from (systematic resampling; the
            // default), or an all-gather of the whole cloud (multinomial resampling, or SMCMI_RESAMPLE_EXCHANGE=allgather)
            static const char *xchg = getenv("SMCMI_RESAMPLE_EXCHANGE");
            const bool a2a = rs && !(xchg && !strcmp(xchg, "allgather")) && rc->resampling_method == SMCMI_RESAMPLE_SYSTEMATIC && g.world > 1;
            if (rs && a2a) {
                // all-to-all-v redistribution: weights are all-gathered (N doubles), every shard forms the global cumulative sum and
                // the ancestor range of every shard's slots; then only the rows inside a shard's range travel to it
                const size_t nloc = (size_t)h0->n;
                if (int rc2 = g.allgather([](smcmi_handle *h) { return (const double *)(h->cl.buf[0] + (long long)(h->R - 1) * h->n); },
                                          [](smcmi_handle *h) { return h->d_full_w; }, nloc)) return rc2;
                for (auto *h : g.hs) {
                    HIP_TRY(hipSetDevice(h->cfg.device));
                    const long long N = h->cfg.n_parts;
                    CloudPtrs wcl{};
                    wcl.buf[0] = wcl.buf[1] = h->d_full_w; wcl.n = N; wcl.R = 1;
                    k_weight_chunk_sums<<<h->nb_full, TB, 0, h->stream>>>(wcl, h->d_st, h->d_part_full);
                    k_chunk_offsets<<<1, 1, 0, h->stream>>>(h->d_st, h->d_part_full, h->nb_full, h->d_off_full, 0.0, 0);
                    k_scan_weights<<<h->nb_full, TB, 0, h->stream>>>(wcl, h->d_st, h->d_off_full, h->d_cum_full, 0, h->nb_full);
                    k_anc_ranges<<<1, 64, 0, h->stream>>>(h->d_st, h->d_cum_full, N, h->n, g.world, h->cfg.seed, h->d_anc);
                }
                std::vector<long long> ranges(2 * (size_t)g.world);
                HIP_TRY(hipSetDevice(h0->cfg.device));
                HIP_TRY(hipMemcpyAsync(ranges.data(), h0->d_anc, sizeof(long long) * ranges.size(), hipMemcpyDeviceToHost, h0->stream));
                HIP_TRY(hipStreamSynchronize(h0->stream));
                if (ranges[0] >= 0) {                      // -1: the device decided not to resample after all
                    if (int rc2 = g.exchange_rows(ranges)) return rc2;
                    for (auto *h : g.hs) {
                        HIP_TRY(hipSetDevice(h->cfg.device));
                        const long long N = h->cfg.n_parts;
                        k_resample_gather<<<(unsigned)((h->n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_st, h->d_cum_full, N, h->cfg.gid0, N,
                                                                                                rc->resampling_method, h->cfg.seed, 0u, nullptr, h->d_anc,
                                                                                                h->d_full_cloud, 0, h->n, 1);
                    }
                }
            } else if (rs) {
                const size_t nloc = (size_t)h0->n;
                if (int rc2 = g.allgather([](smcmi_handle *h) { return (const double *)(h->cl.buf[0] + (long long)(h->R - 1) * h->n); },
                                          [](smcmi_handle *h) { return h->d_full_w; }, nloc)) return rc2;
                if (int rc2 = g.allgather([](smcmi_handle *h) { return (const double *)h->cl.buf[0]; },
                                          [](smcmi_handle *h) { return h->d_full_cloud; }, nloc * h0->R)) return rc2;
                for (auto *h : g.hs) {
                    HIP_TRY(hipSetDevice(h->cfg.device));
                    const long long N = h->cfg.n_parts;
                    CloudPtrs wcl{};
                    wcl.buf[0] = wcl.buf[1] = h->d_full_w; wcl.n = N; wcl.R = 1;
                    k_weight_chunk_sums<<<h->nb_full, TB, 0, h->stream>>>(wcl, h->d_st, h->d_part_full);
                    k_chunk_offsets<<<1, 1, 0, h->stream>>>(h->d_st, h->d_part_full, h->nb_full, h->d_off_full, 0.0, 0);
                    k_scan_weights<<<h->nb_full, TB, 0, h->stream>>>(wcl, h->d_st, h->d_off_full, h->d_cum_full, 0, h->nb_full);
                    k_resample_gather<<<(unsigned)((h->n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_st, h->d_cum_full, N, h->cfg.gid0, N,
                                                                                            rc->resampling_method, h->cfg.seed, 0u, nullptr, h->d_anc,
                                                                                            h->d_full_cloud, 0, h->n, 1);
                }
            }
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                const int nbm = launch_moments(h, h->d_hist_W, 0);
                k_moments_reduce<<<(h->npairs + 63) / 64, 1024, 0, h->stream>>>(h->d_st, h->d_part_mom, nbm, h->npairs, h->d_tot_mom, 0);
            }
        }
        if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_mom; }, h0->npairs)) return rc2;
        if (host_mut) {
            // the closure scores every shard's proposals; a stage that did not run on the device (stall, pause, ϕ = 1 reached, error - the
            // same on every rank: all decisions come from all-reduced totals) must not reach it
            for (auto *h : g.hs) { HIP_TRY(hipSetDevice(h->cfg.device)); launch_prepare_in_run(h, h->d_tot_mom, 1, 2); }
            int dn = 0;
            HIP_TRY(hipSetDevice(h0->cfg.device));
            HIP_TRY(hipMemcpyAsync(&dn, &h0->d_st->done, sizeof(int), hipMemcpyDeviceToHost, h0->stream));
            HIP_TRY(hipStreamSynchronize(h0->stream));
            if (dn) return 0;
            // (a closure that fails on ONE rank must end the stage on every rank: the failure rides the acceptance all-reduce as a second
            // number - a rank that returned alone would leave the others waiting in the collective)
            int my_err = 0;
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                const int e = my_err ? my_err : host_mutation(h, rc, tempered_cb);
                if (e) my_err = e;
                else k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_acc_part, h->nb_mut, 1, h->d_tot_acc + EACC);
                const double flag = e ? 1.0 : 0.0;
                HIP_TRY(hipMemcpyAsync(h->d_tot_acc + EACC + 1, &flag, sizeof(double), hipMemcpyHostToDevice, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));           // (`flag` is a stack variable)
            }
            const std::string my_msg = my_err ? g_err : std::string();
            if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_acc + EACC; }, 2)) return rc2;
            double failed = 0.0;
            HIP_TRY(hipSetDevice(h0->cfg.device));
            HIP_TRY(hipMemcpyAsync(&failed, h0->d_tot_acc + EACC + 1, sizeof(double), hipMemcpyDeviceToHost, h0->stream));
            HIP_TRY(hipStreamSynchronize(h0->stream));
            if (my_err) return set_err(my_err, my_msg);
            if (failed != 0.0) return set_err(SMCMI_ERR_CALLBACK, "the likelihood callback failed on another rank of the sharded run");
            return 0;
        }
        for (auto *h : g.hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            launch_prepare_in_run(h, h->d_tot_mom, 1, 2);
            hipEvent_t pe0 = nullptr, pe1 = nullptr;
            if (profile && h == h0) { hipEventCreate(&pe0); hipEventCreate(&pe1); mut_evs.push_back(pe0); mut_evs.push_back(pe1); hipEventRecord(pe0, h->stream); }
            const int nbl = launch_mutate(h, rc->n_blocks, 0, rc->alpha);
            if (pe1) hipEventRecord(pe1, h->stream);
            if (predict) k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_esum_part, nbl, ES, h->d_tot_acc, h->d_emax_part, nbl, h->d_tot_acc + ES, shard_rank(h), g.world);
            else k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_acc_part, nbl, 1, h->d_tot_acc + EACC, h->d_emax_part, nbl, h->d_tot_acc + ES, shard_rank(h), g.world);
        }
        if (predict) { if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_acc; }, ES + g.world)) return rc2; }
        else if (int rc2 = g.allreduce([](smcmi_handle *h) { return h->d_tot_acc + EACC; }, 1 + g.world)) return rc2;
        return 0;
    };

    const auto t0 = std::chrono::steady_clock::now();
    const bool cont = rc->continue_run != 0;
    const int base = cont ? h0->h_st.stage - 1 : 0;      // stages completed before this call
    const int max_iter = (adaptive ? h0->cfg.max_stages : rc->n_phi - 1) - base;
    const int sync_every = rc->sync_every > 0 ? rc->sync_every : 16;
    const int first_passes = std::max(P_default, FIRST_SOLVER_PASSES);
    const double N_tot = (double)h0->cfg.n_parts, thr = rc->threshold_ratio * N_tot;
    double pred_ess = cont ? h0->h_st.ess_prev : (rc->initial_ess > 0.0 ? rc->initial_ess : N_tot);
    int pred_rl = cont ? h0->h_st.resampled_last : 0, stall_stage = -1, stall_p = 0;
    int done = 0, iters = 0, stalls = 0, sel_stalls = 0, spec_stalls = 0;
    const bool spec_ok = predict_select && predict && can_fuse_cm(h0) &&
                         rc->tempered_update_prior_weight == 0.0 && !(rc->phi_rtol < 0.0);
    bool spec_on = spec_ok;                   // switched off after repeated verification failures (see smcmi_run)
    int last_spec_stall = -100, spec_strikes = 0, last_solver_stall = -100;
    int dyn_P = P_default;                    // raised when stages keep running out of passes
    int stages_left_est = 1 << 30;            // (1 - ϕ_n) / (ϕ_n - ϕ_{n-1}) at the last sync (smcmi_run)
    while (iters < max_iter && !done) {
        const int batch = host_mut ? 1 : (adaptive ? std::min(std::min(sync_every, std::max(stages_left_est, 4)), max_iter - iters) : max_iter - iters);
        for (int b = 0; b < batch; ++b) {
            int mode = 0;
            if (predict_select) {
                const double ess_bar = rc->tempering_target * (pred_rl ? N_tot : pred_ess);
                const bool rs = ess_bar < thr * (1.0 + 1e-6);
                mode = (!rs || sel_mode == 2) ? 1 : 0;
                pred_ess = ess_bar; pred_rl = rs ? 1 : 0;
            }
            const bool spec = spec_on && mode == 1 && iters >= 2;
            if (int rc2 = enqueue(0, adaptive ? (iters < 2 ? first_passes : dyn_P) : 0, mode, spec)) return rc2;
            ++iters;
        }
        // one 64-byte copy per sync: done flag, resample flag and ESS of the last stage are contiguous in DevState (smcmi_run)
        DevState head;
        constexpr size_t head_off = offsetof(DevState, stage), head_len = offsetof(DevState, ess) - offsetof(DevState, stage);
        for (;;) {
            HIP_TRY(hipSetDevice(h0->cfg.device));
            HIP_TRY(hipMemcpyAsync((char *)&head + head_off, (const char *)h0->d_st + head_off, head_len, hipMemcpyDeviceToHost, h0->stream));
            HIP_TRY(hipStreamSynchronize(h0->stream));
            done = head.done;
            if (done != 2 && done != 3 && done != 4) break;
            // stall (identical on every rank: all decisions come from all-reduced totals): clear it and resume that stage
            if (pull_state(h0)) return SMCMI_ERR_HIP;
            const int st_i = h0->h_st.stage;
            const int had = (st_i == stall_stage) ? stall_p : (st_i - base <= 3 ? first_passes : dyn_P);
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                const int zero = 0;
                HIP_TRY(hipMemcpyAsync(&h->d_st->done, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
            }
            if (done == 4) {
                // a stage without certificate pass had no usable / verified prediction: nothing is committed, redo it in full
                for (auto *h : g.hs) {
                    HIP_TRY(hipSetDevice(h->cfg.device));
                    k_solver_rearm<<<1, 64, 0, h->stream>>>(h->d_st, h->d_sched);
                }
                if (int rc2 = enqueue(0, first_passes, 0, false, true)) return rc2;
                stall_stage = st_i; stall_p = first_passes;
                ++spec_stalls;
                if (st_i - last_spec_stall <= 4) { if (++spec_strikes >= 2) spec_on = false; }
                else spec_strikes = 0;
                last_spec_stall = st_i;
            } else if (done == 2) {
                if (int rc2 = enqueue(had, had + 4, 0)) return rc2;
                stall_stage = st_i; stall_p = had + 4;
                ++stalls;
                if (st_i - last_solver_stall <= 4 && dyn_P < 4) ++dyn_P;
                last_solver_stall = st_i;
            } else {
                if (int rc2 = enqueue(had, had, 2)) return rc2;
                ++sel_stalls;
            }
            iters = st_i - 1 - base;
        }
        if (!done && head.phi_n >= 1.0) break;      // the last stage of the batch reached ϕ = 1: the closing k_stage_begin finishes (smcmi_run)
        if (head.phi_n > head.phi_prev && head.phi_n < 1.0) {
            const double left = (1.0 - head.phi_n) / (head.phi_n - head.phi_prev);
            stages_left_est = left < 1e6 ? (int)left + 1 : 1 << 30;
        }
        if (predict_select) { pred_rl = head.resampled_last; pred_ess = head.ess_prev; }    // (the copy that ended the loop above)
    }
    // fold the last acceptance rate, close the run
    for (auto *h : g.hs) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        k_stage_begin<<<1, BT, 0, h->stream>>>(h->d_st, h->d_sched, h->d_tot_acc + EACC, 1, h->rec);
        if (pull_state(h)) return SMCMI_ERR_HIP;
        h->last_n_stages = h->h_st.stage;
    }
    const auto t1 = std::chrono::steady_clock::now();
    const DevState &s = h0->h_st;
    memset(res, 0, sizeof(*res));
    res->n_stages = s.stage; res->resamples = s.resamples; res->logmdd = s.logz; res->c = s.c; res->accept = s.accept;
    res->seconds = std::chrono::duration<double>(t1 - t0).count();
    res->solver_passes = s.solver_passes;
    res->solver_stalls = stalls; res->select_stalls = sel_stalls; res->spec_stalls = spec_stalls;
    res->paused = (s.done == 5) ? 1 : 0;
    if (!mut_evs.empty()) {
        // event pairs bracket dispatch + kernel; launches of stalled (no-op) stages are short and rare - they stay in the mean.
        // The dispatch part is calibrated like in smcmi_run: pairs around an empty kernel of the same grid, minus its own ~2.5 µs.
        HIP_TRY(hipSetDevice(h0->cfg.device));
        hipEvent_t c0, c1;
        hipEventCreate(&c0); hipEventCreate(&c1);
        double acc_ms = 0.0;
        int got = 0;
        for (int r = 0; r < 64; ++r) {
            k_fill<<<(unsigned)((h0->n + 255) / 256), 256, 0, h0->stream>>>(nullptr, 0, 0.0);      // (an empty launch of the mutation kernel's grid: event-overhead calibration)
            hipEventRecord(c0, h0->stream);
            k_fill<<<(unsigned)((h0->n + 255) / 256), 256, 0, h0->stream>>>(nullptr, 0, 0.0);      // (an empty launch of the mutation kernel's grid: event-overhead calibration)
            hipEventRecord(c1, h0->stream);
            hipStreamSynchronize(h0->stream);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c0, c1) == hipSuccess) { acc_ms += ms; ++got; }
        }
        hipEventDestroy(c0); hipEventDestroy(c1);
        const double over = got ? std::max(0.0, acc_ms / got - 0.0025) : 0.0;
        for (size_t k = 0; k + 1 < mut_evs.size(); k += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, mut_evs[k], mut_evs[k + 1]) == hipSuccess) { res->kernel_ms_mutate += std::max(0.0, (double)ms - over); res->n_mutate_launches += 1; }
        }
        for (hipEvent_t e : mut_evs) hipEventDestroy(e);
    }
    if (s.err) return err_from_state(s.err);
    if (!s.done) return set_err(SMCMI_ERR_CAPACITY, "max_stages exceeded before the tempering schedule reached 1");
    return 0;
}

extern "C" int smcmi_run_sharded(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res) {
    if (int e = need_model(h, 2)) return e;
    if (!rc || !res) return set_err(SMCMI_ERR_ARG, "null argument");
    res->n_segments = 0; res->segment_stages = 0; res->kernel_ms_segments = 0.0;
    res->segment_blocks = 0; res->segment_state = 0; res->segment_timeouts = 0; res->shift_fallback_stage = 0;
    if (!h->nccl && !h->has_hostc) return set_err(SMCMI_ERR_STATE, "smcmi_comm_init / smcmi_comm_init_host has not been called on this handle");
    if (int e = check_lik_pair(h)) return e;
    ShardGroup g;
    g.hs = {h}; g.world = h->world; g.rccl = true; g.hostc = h->has_hostc;
    if (!h->cb[0] && eng2_eligible(h, g.world, false, rc)) return run2_guarded(g, rc, res);      // n_para <= 10: the two-launch stage (stage2.hpp / run2.hpp)
    return run_sharded_impl(g, rc, res);                                             // n_para > 10, and every run with a host likelihood
}

extern "C" int smcmi_run_group(smcmi_handle **hs, int32_t n, const smcmi_run_config *rc, smcmi_result *res) {
    if (!hs || n < 1 || !rc || !res) return set_err(SMCMI_ERR_ARG, "bad argument");
    res->n_segments = 0; res->segment_stages = 0; res->kernel_ms_segments = 0.0;
    res->segment_blocks = 0; res->segment_state = 0; res->segment_timeouts = 0; res->shift_fallback_stage = 0;
    ShardGroup g;
    long long expect = 0;
    for (int k = 0; k < n; ++k) {
        if (int e = need_model(hs[k], 2)) return e;
        if (int e = check_lik_pair(hs[k])) return e;
        if (hs[k]->cfg.gid0 != expect || hs[k]->cfg.n_local != hs[0]->cfg.n_local || hs[k]->cfg.n_parts != hs[0]->cfg.n_parts)
            return set_err(SMCMI_ERR_ARG, "group handles must be equal contiguous shards in rank order");
        expect += hs[k]->cfg.n_local;
        g.hs.push_back(hs[k]);
    }
    if (expect != hs[0]->cfg.n_parts) return set_err(SMCMI_ERR_ARG, "group handles do not cover n_parts");
    g.world = n; g.rccl = false;
    if (!hs[0]->cb[0] && eng2_eligible(hs[0], g.world, n == 1, rc)) return run2_guarded(g, rc, res);
    return run_sharded_impl(g, rc, res);
}
