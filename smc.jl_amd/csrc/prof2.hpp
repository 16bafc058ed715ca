// prof2.hpp - development only (SMCMI_PROF2=<stage>): what the stamps of that stage say - worker 0's phases inside a persistent segment, the two
// hand-overs on the wall clock all dies share, the block census of a large-shard mutation launch, K1 / K2 block ticks.  Included by run2.hpp.
#pragma once

static int prof2_report(smcmi_handle *h0, const Geo2 &g0, int seg_launches) {
    {
        long long pr[128], gt[64], gp[64];
        HIP_TRY(hipMemcpy(pr, h0->e2->d_prof + PROF2_SEG0, 64 * sizeof(long long), hipMemcpyDeviceToHost));      // worker 0 of a segment
        HIP_TRY(hipMemcpy(gt, h0->e2->d_prof + PROF2_GATH, sizeof(gt), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(gp, h0->e2->d_prof + PROF2_GPOLL, sizeof(gp), hipMemcpyDeviceToHost));
        if (seg_launches > 0) {
            // worker block 0's phases of stage prof_stage and the decider's (its clock has another origin: only its own differences mean anything)
            fprintf(stderr, "[smcmi3] stage %d worker 0 ticks: correction + row %lld | wait for the totals %lld | decision + proposal %lld | MH steps %lld | mutation row %lld | next stage's draws %lld | wait for the totals %lld | begin %lld | stage %lld\n",
                    h0->e2->prof_stage, pr[2] - pr[1], pr[3] - pr[2], pr[4] - pr[3], pr[5] - pr[4], pr[6] - pr[5], pr[7] - pr[6], pr[8] - pr[7], pr[9] - pr[8], pr[9] - pr[1]);
            // (a riding launch takes the steps in another order - CORR (2), DRAW (7), BEGIN (8, 9), totals (3), proposal (4), MH (5), row (6): the stamps themselves)
            fprintf(stderr, "[smcmi3]   stamps relative to the stage's first:");
            for (int q = 1; q <= 9; ++q) fprintf(stderr, " [%d] %lld", q, pr[q] - pr[1]);
            fprintf(stderr, "\n");
            {
                long long sp[10];
                HIP_TRY(hipMemcpy(sp, h0->e2->d_prof + PROF2_SEL, sizeof(sp), hipMemcpyDeviceToHost));
                if (sp[9]) fprintf(stderr, "[smcmi3]   selection inside the segment (wall clock, us): particle stored %.2f | chunk offsets %.2f | scan + cum %.2f | stores acknowledged %.2f | hand-over %.2f | chunk ends %.2f | search %.2f | rows gathered %.2f | moment row %.2f | hand-over %.2f\n",
                                   0.0, (sp[1] - sp[0]) * 0.01, (sp[2] - sp[1]) * 0.01, (sp[3] - sp[2]) * 0.01, (sp[4] - sp[3]) * 0.01, (sp[5] - sp[4]) * 0.01, (sp[6] - sp[5]) * 0.01, (sp[7] - sp[6]) * 0.01, (sp[8] - sp[7]) * 0.01, (sp[9] - sp[8]) * 0.01);
            }
            fprintf(stderr, "[smcmi3]   decision + proposal: totals -> covariance, shuffle %lld | block matrices %lld | Cholesky + log det %lld | rest %lld\n",
                    pr[30] - pr[3], pr[31] - pr[30], pr[32] - pr[31], pr[4] - pr[32]);
        }
        if (seg_launches > 0) {
            // the two hand-overs of that stage on the wall clock (10 ns ticks): when the workers' rows went out, what the gatherers did, when
            // the workers had the totals
            const int Wk = g0.Vl * g0.nb2;
            std::vector<long long> ws(4 * (size_t)Wk);
            HIP_TRY(hipMemcpy(ws.data(), h0->e2->d_prof + PROF2_WORK, sizeof(long long) * ws.size(), hipMemcpyDeviceToHost));
            for (int kind = 0; kind < 2; ++kind) {
                long long p_min = 0, p_max = 0, s_min = 0, s_max = 0;
                for (int b = 0; b < Wk; ++b) {
                    const long long pb = ws[4 * b + 2 * kind], sb = ws[4 * b + 2 * kind + 1];
                    if (!b || pb < p_min) p_min = pb;
                    if (!b || pb > p_max) p_max = pb;
                    if (!b || sb < s_min) s_min = sb;
                    if (!b || sb > s_max) s_max = sb;
                }
                fprintf(stderr, "[smcmi3] hand-over %d (%s rows): rows published over %.2f us (worker 0 at +%.2f); totals seen by the first worker +%.2f, the last +%.2f, worker 0 +%.2f after the first row\n",
                        kind, kind ? "mutation" : "correction", (p_max - p_min) * 0.01, (ws[2 * kind] - p_min) * 0.01, (s_min - p_min) * 0.01, (s_max - p_min) * 0.01, (ws[2 * kind + 1] - p_min) * 0.01);
                for (int v = 0; v < g0.Vl; ++v) {
                    long long lastrow = 0;
                    for (int b = v; b < Wk; b += g0.Vl) lastrow = std::max(lastrow, ws[4 * b + 2 * kind]);
                    fprintf(stderr, "[smcmi3]   gatherer %d: its last row +%.2f | first words seen +%.2f | %lld sweep(s) done +%.2f | totals posted +%.2f (started waiting at +%.2f)\n", v, (lastrow - p_min) * 0.01,
                            (gp[4 * v + 2 * kind] - p_min) * 0.01, gp[4 * v + 2 * kind + 1],
                            (gt[6 * v + 3 * kind + 1] - p_min) * 0.01, (gt[6 * v + 3 * kind + 2] - p_min) * 0.01, (gt[6 * v + 3 * kind] - p_min) * 0.01);
                }
            }
        }
        if (!g0.direct && !g0.inker && !g0.wide) {
            // census of the large-shard mutation launch of the profiled stage (100 MHz wall clock, the CU every block sat on): how many blocks
            // a CU held at once
            std::vector<long long> cs(3 * PROF2_BLOCKS);
            HIP_TRY(hipMemcpy(cs.data(), h0->e2->d_prof + PROF2_CENSUS, sizeof(long long) * cs.size(), hipMemcpyDeviceToHost));
            long long t_min = 0, t_max = 0, sum = 0;
            int nbk = 0;
            std::map<long long, std::vector<std::pair<long long, int>>> per_cu;
            for (int b = 0; b < PROF2_BLOCKS; ++b) {
                const long long t_a = cs[3 * b], t_b = cs[3 * b + 1];
                if (!t_a || !t_b) continue;
                if (!nbk || t_a < t_min) t_min = t_a;
                if (!nbk || t_b > t_max) t_max = t_b;
                sum += t_b - t_a;
                ++nbk;
                per_cu[cs[3 * b + 2]].push_back({t_a, +1});
                per_cu[cs[3 * b + 2]].push_back({t_b, -1});
            }
            int hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (auto &kv : per_cu) {
                std::sort(kv.second.begin(), kv.second.end());
                int cur = 0, mx = 0;
                for (auto &ev : kv.second) { cur += ev.second; mx = std::max(mx, cur); }
                hist[std::min(mx, 8)] += 1;
            }
            if (nbk) {
                fprintf(stderr, "[smcmi2] K2b census: %d blocks on %d CUs, first start -> last end %.2f us, mean block %.2f us, mean residency %.1f blocks; CUs by the most blocks they held at once:", nbk,
                        (int)per_cu.size(), (t_max - t_min) * 0.01, sum * 0.01 / nbk, (double)sum / (double)std::max<long long>(1, t_max - t_min));
                for (int k = 1; k <= 8; ++k) if (hist[k]) fprintf(stderr, " %d x %d", hist[k], k);
                // when the blocks started (µs behind the first) and how long they ran
                int st_h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, du_h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const double st_e[7] = {1, 2, 5, 10, 15, 20, 25}, du_e[7] = {8, 12, 16, 20, 24, 28, 32};
                for (int b = 0; b < PROF2_BLOCKS; ++b) {
                    if (!cs[3 * b] || !cs[3 * b + 1]) continue;
                    const double st = (cs[3 * b] - t_min) * 0.01, du = (cs[3 * b + 1] - cs[3 * b]) * 0.01;
                    int k = 0; while (k < 7 && st >= st_e[k]) ++k; st_h[k] += 1;
                    k = 0; while (k < 7 && du >= du_e[k]) ++k; du_h[k] += 1;
                }
                fprintf(stderr, "\n[smcmi2]   started at <1 <2 <5 <10 <15 <20 <25 >=25 us:");
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", st_h[k]);
                fprintf(stderr, "; ran <8 <12 <16 <20 <24 <28 <32 >=32 us:");
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", du_h[k]);
                fprintf(stderr, "\n");
            }
        }
        HIP_TRY(hipMemcpy(pr, h0->e2->d_prof + PROF2_K1, 128 * sizeof(long long), hipMemcpyDeviceToHost));          // K1's and K2's stamps
        for (int blk = 0; blk < 2; ++blk) {
            fprintf(stderr, "[smcmi2] K1 %s block ticks:", blk ? "mid" : "0");
            for (int q = 1; q <= 5; ++q) fprintf(stderr, " %lld", pr[blk * 32 + q] - pr[blk * 32 + q - 1]);
            fprintf(stderr, "  total %lld\n[smcmi2] K2 %s block ticks:", pr[blk * 32 + 5] - pr[blk * 32], blk ? "mid" : "0");
            for (int q = 1; q <= 11; ++q) fprintf(stderr, " %lld", pr[64 + blk * 32 + q] - pr[64 + blk * 32 + q - 1]);
            fprintf(stderr, "  total %lld\n", pr[64 + blk * 32 + 10] - pr[64 + blk * 32]);
        }
    }
    return 0;
}
