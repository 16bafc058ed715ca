// mailbox.hpp - the peer mailbox's host side (stage2.hpp has the device side and the wire format): allocation of a handle's tables in fine-grained
// memory, the table of peer addresses, HIP IPC export / import between processes, the transport's self-test, and the set-up through whichever
// communicator a sharded run has (sharded.hpp ShardGroup).  Included by run2.hpp.
#pragma once

// ---- peer mailbox (stage2.hpp): allocation and the table of peer addresses
// clear the sticky time-out flag and (re)load the time-out (SMCMI_MAILBOX_TIMEOUT_MS, default 10 s; read at every run)
static int mbox_reset_flag(smcmi_handle *h, hipStream_t s) {
    const char *ms = getenv("SMCMI_MAILBOX_TIMEOUT_MS");
    const unsigned long long fl[MB_FLAG_WORDS] = {0ull, ms && atof(ms) > 0.0 ? (unsigned long long)(atof(ms) * 1e5) : (unsigned long long)MB_TIMEOUT_TICKS_DEFAULT};
    if (s) { HIP_TRY(hipMemcpyAsync(h->d_mbox + MB_WORDS, fl, sizeof(fl), hipMemcpyHostToDevice, s)); HIP_TRY(hipStreamSynchronize(s)); }
    else HIP_TRY(hipMemcpy(h->d_mbox + MB_WORDS, fl, sizeof(fl), hipMemcpyHostToDevice));
    return 0;
}
// words behind the tables for the selection inside sharded segments (stage2.hpp MB_SEL_OFF): only handles whose shard can run segments
// (n_para <= 10, at most 131 072 particles); every handle of a run has the same (N, n, n_para), hence the same layout
static size_t mbox_sel_words(const smcmi_handle *h) {
    if (h->d > 10 || h->n > 131072 || h->cfg.n_parts > (long long)V2_MAXV * 65536) return 0;      // (segments: correction rows of 512 particles, at most 128 per virtual shard)
    return (size_t)MB_SEL_TABLE_WORDS + (size_t)h->cfg.n_parts + (size_t)(h->d + 4) * (size_t)h->n;
}
static int mbox_alloc(smcmi_handle *h) {
    if (h->d_mbox) return 0;
    HIP_TRY(hipSetDevice(h->cfg.device));
    // fine-grained: stores from a peer GPU and this GPU's polling loads meet in memory, not in a die's L2
    const size_t words = (size_t)MB_ALLOC_WORDS + mbox_sel_words(h);
    HIP_TRY(hipExtMallocWithFlags((void **)&h->d_mbox, sizeof(unsigned long long) * words, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemset(h->d_mbox, 0xFF, sizeof(unsigned long long) * words));
    if (int e = mbox_reset_flag(h, nullptr)) return e;
    HIP_TRY(hipDeviceSynchronize());              // (null-stream fill: not ordered with the handle's non-blocking stream)
    return 0;
}
static int mbox_set_peers(smcmi_handle *h, const std::vector<unsigned long long *> &peers) {
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->d_peers) { hipFree(h->d_peers); h->d_peers = nullptr; }
    HIP_TRY(hipMalloc((void **)&h->d_peers, sizeof(unsigned long long *) * peers.size()));
    HIP_TRY(hipMemcpy(h->d_peers, peers.data(), sizeof(unsigned long long *) * peers.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    h->h_peers = peers;
    return 0;
}
// several handles of one process (tests; SMCMI_MAILBOX=1): every handle sees the others' tables directly
static int mbox_setup_group(ShardGroup &g) {
    std::vector<unsigned long long *> peers(g.hs.size());
    for (auto *h : g.hs) {
        if (int e = mbox_alloc(h)) return e;
        peers[shard_rank(h)] = h->d_mbox;
    }
    for (auto *h : g.hs)
        if (h->h_peers != peers) { if (int e = mbox_set_peers(h, peers)) return e; }
    return 0;
}
// ---- peer mailbox across processes: HIP IPC handles of the tables, exchanged by the caller or through the communicator
// One block: `rounds` exchanges of a (rank, round)-dependent row with every peer over the real transport; errs += mismatches / time-outs.
static __global__ void k_mbox_selftest(unsigned long long *const *peers, const unsigned long long *mine, int world, int rank, int rounds, int *errs) {
    const int t = threadIdx.x;
    int bad = 0;
    for (int q = 0; q < rounds; ++q) {
        const unsigned tag = 0x7F000000u | (unsigned)q;
        const long long table = (long long)(q & 1) * MB_TABLE_WORDS;
        if (t < 16)
            for (int r = 0; r < world; ++r) mb_store(peers[r] + table + ((long long)rank * MB_LD + t) * 2, 1000.0 * rank + q + t / 16.0, tag);
        for (int idx = t; idx < world * 16; idx += blockDim.x) {
            const int r = idx / 16, k = idx % 16;
            const double x = mb_load(mine + table + ((long long)r * MB_LD + k) * 2, tag, const_cast<unsigned long long *>(mine) + MB_WORDS);
            if (!(x == 1000.0 * r + q + k / 16.0)) ++bad;
        }
        __syncthreads();                     // (a rank re-uses a table two rounds later: only after it has read it)
    }
    if (bad) atomicAdd(errs, bad);
}
static void mbox_close_peers(smcmi_handle *h) {
    for (void *p : h->ipc_opened) hipIpcCloseMemHandle(p);
    h->ipc_opened.clear();
    h->h_peers.clear();
}
static int mbox_export(smcmi_handle *h, uint8_t *out64) {
    if (int e = mbox_alloc(h)) return e;
    hipIpcMemHandle_t hd;
    HIP_TRY(hipIpcGetMemHandle(&hd, h->d_mbox));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
    memcpy(out64, &hd, 64);
    return 0;
}
static int mbox_import(smcmi_handle *h, int world, int rank, const uint8_t *all) {
    if (world < 1 || world > V2_MAXV || rank < 0 || rank >= world) return set_err(SMCMI_ERR_ARG, "mailbox: bad (rank, world)");
    if (int e = mbox_alloc(h)) return e;
    HIP_TRY(hipSetDevice(h->cfg.device));
    mbox_close_peers(h);
    std::vector<unsigned long long *> peers((size_t)world, nullptr);
    for (int r = 0; r < world; ++r) {
        if (r == rank) { peers[r] = h->d_mbox; continue; }
        hipIpcMemHandle_t hd;
        memcpy(&hd, all + 64 * (size_t)r, 64);
        void *p = nullptr;
        if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) {
            (void)hipGetLastError();
            mbox_close_peers(h);
            return set_err(SMCMI_ERR_HIP, "mailbox: hipIpcOpenMemHandle failed for rank " + std::to_string(r));
        }
        h->ipc_opened.push_back(p);
        peers[r] = (unsigned long long *)p;
    }
    return mbox_set_peers(h, peers);
}
// errors (mismatches + time-outs) of `rounds` exchanges with every peer; every rank must call it at the same time
static int mbox_selftest(smcmi_handle *h, int world, int rank, int rounds, int *errs_out) {
    if (!h->d_peers || (int)h->h_peers.size() != world) return set_err(SMCMI_ERR_STATE, "mailbox: peers not imported");
    HIP_TRY(hipSetDevice(h->cfg.device));
    int *d_err = nullptr;
    HIP_TRY(hipMalloc((void **)&d_err, sizeof(int)));
    HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int), h->stream));
    k_mbox_selftest<<<1, 128, 0, h->stream>>>(h->d_peers, h->d_mbox, world, rank, rounds, d_err);
    int e = 0;
    HIP_TRY(hipMemcpyAsync(&e, d_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    hipFree(d_err);
    *errs_out = e;
    return 0;
}
// One handle per process (RCCL or the host-mediated communicator): map every rank's table through the communicator and test the
// transport; all ranks reach the same verdict (h->mbox_ok) - anything short of a clean self-test on every rank leaves the all-gathers
// in place.
static int mbox_setup_remote(ShardGroup &g) {
    smcmi_handle *h = g.hs[0];
    if (h->mbox_tried) return 0;
    h->mbox_tried = true; h->mbox_ok = false;
    if (h->world > V2_MAXV) return 0;
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int world = h->world, rank = h->rank;
    uint8_t mine[64] = {0};
    double fail = mbox_export(h, mine) ? 1.0 : 0.0;
    double *d_send = nullptr, *d_recv = nullptr;
    HIP_TRY(hipMalloc((void **)&d_send, 64));
    HIP_TRY(hipMalloc((void **)&d_recv, 64 * (size_t)world));
    HIP_TRY(hipMemcpyAsync(d_send, mine, 64, hipMemcpyHostToDevice, h->stream));
    if (int e = g.allgather([=](smcmi_handle *) { return (const double *)d_send; }, [=](smcmi_handle *) { return d_recv; }, (size_t)8)) {   // 64 bytes = 8 doubles per rank
        hipFree(d_send); hipFree(d_recv);
        return e;
    }
    std::vector<uint8_t> all(64 * (size_t)world);
    HIP_TRY(hipMemcpyAsync(all.data(), d_recv, all.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    hipFree(d_send); hipFree(d_recv);
    if (fail == 0.0 && mbox_import(h, world, rank, all.data())) fail = 1.0;
    auto agree = [&](double mine_bad, double *total) -> int {           // sum of the ranks' failure counts
        HIP_TRY(hipMemcpyAsync(h->d_comm, &mine_bad, sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (int e = g.allreduce([](smcmi_handle *hh) { return hh->d_comm; }, 1)) return e;
        HIP_TRY(hipMemcpyAsync(total, h->d_comm, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return 0;
    };
    double total = 0.0;
    if (int e = agree(fail, &total)) return e;
    if (total != 0.0) { mbox_close_peers(h); return 0; }               // some rank could not map: everybody keeps the all-gathers
    int errs = 0;
    if (mbox_selftest(h, world, rank, 256, &errs)) errs = 1;
    if (int e = agree((double)errs, &total)) return e;
    if (total != 0.0) { mbox_close_peers(h); return 0; }
    h->mbox_ok = true;
    return 0;
}
static long long mbox_table(int kind, unsigned cnt) { return (long long)(kind * 2 + (int)(cnt & 1u)) * MB_TABLE_WORDS; }
static unsigned mbox_tag(unsigned epoch, unsigned cnt) { return ((epoch & 0x7Fu) << 24) | (cnt & 0xFFFFFFu); }      // (never 0xFFFFFFFF: a cleared word)

