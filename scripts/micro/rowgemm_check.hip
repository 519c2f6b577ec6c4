// Correctness harness of k_train_bwd_dx / k_train_fwd against a CPU loop on random data (debug aid for kernels_train_big.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include "../../gnnkeras_amd/csrc/kernels_train_big.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <typename T> T *up(const std::vector<T> &v) { T *d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

template <int HQ, int NCT, bool B6 = false>
void check_bwd(int M, bool bn, bool scale) {
    const int S = 8 * NCT, H = 16 * HQ, L = 7, K = 2 * S + 2 * L + 3, wa = S + L;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0, 1);
    std::vector<float> dZ((size_t)M * H), W((size_t)K * H), st((size_t)M * S), ag((size_t)M * S), gm(K), mu(K), va(K), m1(K), m2(K), rs(M);
    for (auto *v : {&dZ, &W, &st, &ag, &gm, &mu, &m1, &m2}) for (auto &x : *v) x = nd(rng);
    for (auto &x : va) x = fabsf(nd(rng)) + 0.1f;
    for (auto &x : rs) x = 0.5f + fabsf(nd(rng));
    gnn::TrainBwdArgs a; memset(&a, 0, sizeof(a));
    float *d_dx; CK(hipMalloc(&d_dx, (size_t)M * 2 * S * 4)); CK(hipMemset(d_dx, 0xff, (size_t)M * 2 * S * 4));
    a.M = M; a.dZ = up(dZ); a.ldz = H; a.W = up(W); a.ldw = H; a.H = H; a.S = S; a.wrow_state = 0; a.wrow_agg = wa;
    a.state = up(st); a.ld_state = S; a.agg = up(ag); a.ld_agg = S;
    if (bn) { a.gamma = up(gm); a.mean = up(mu); a.var = up(va); a.m1 = up(m1); a.m2 = up(m2); a.eps = 1e-3f; }
    if (scale) a.agg_row_scale = up(rs);
    a.dx = d_dx; a.ld_dx = 2 * S;
    if constexpr (B6) gnn::k_train_bwd_dx_b6<HQ, GNN_ACT_LINEAR><<<64, 256, gnn::train_bwd_b6_lds<HQ>()>>>(a);
    else gnn::k_train_bwd_dx<HQ, NCT><<<64, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<HQ, NCT>()>>>(a);
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)M * 2 * S);
    CK(hipMemcpy(got.data(), d_dx, got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < M; ++m) for (int j = 0; j < 2 * S; ++j) {
        const int row = j < S ? j : wa + (j - S);
        double dy = 0; for (int h = 0; h < H; ++h) dy += (double)dZ[(size_t)m * H + h] * W[(size_t)row * H + h];
        double v = dy;
        if (bn) { const double rstd = 1.0 / sqrt(va[row] + 1e-3), x = j < S ? st[(size_t)m * S + j] : ag[(size_t)m * S + j - S];
                  v = gm[row] * rstd * (dy - m1[row] - (x - mu[row]) * rstd * m2[row]); }
        if (scale && j >= S) v *= rs[m];
        worst = fmax(worst, fabs(v - got[(size_t)m * 2 * S + j]));
    }
    printf("bwd<%d,%d>%s M=%d bn=%d scale=%d  max abs err %.3e\n", HQ, NCT, B6 ? " bf16x6" : "", M, bn, scale, worst);
}

template <int SQ, int W32 = 0>
void check_fwd(int M, bool pred, bool centred = false) {
    const int S = 16 * SQ, L = 7, A_ = 3, K = 2 * S + 2 * L + A_, wa = S + L;
    std::mt19937 rng(2); std::normal_distribution<float> nd(0, 1);
    std::vector<float> st((size_t)M * S), ag((size_t)M * S), xc((size_t)M * 32, 0.f), Wf((size_t)K * S), bf(S);
    for (auto *v : {&st, &ag, &Wf, &bf}) for (auto &x : *v) x = 0.3f * nd(rng);
    for (int m = 0; m < M; ++m) { for (int j = 0; j < 2 * L + A_; ++j) xc[(size_t)m * 32 + j] = nd(rng); xc[(size_t)m * 32 + 2 * L + A_] = 1.f; }
    gnn::TrainFwdArgs a; memset(&a, 0, sizeof(a));
    float *dY, *part; int *flag;
    CK(hipMalloc(&dY, (size_t)M * S * 4)); CK(hipMemset(dY, 0xff, (size_t)M * S * 4)); CK(hipMalloc(&part, 64 * 2 * S * 4)); CK(hipMalloc(&flag, 8)); CK(hipMemset(flag, 0, 8));
    a.M = M; a.state = up(st); a.ld_state = S; a.agg = up(ag); a.ld_agg = S; a.xc = up(xc); a.Wf = up(Wf); a.bf = up(bf); a.H = S;
    a.wrow_state = 0; a.wrow_agg = wa; a.cs.n = 3; a.cs.width[0] = L; a.cs.wrow[0] = S; a.cs.width[1] = L; a.cs.wrow[1] = 2 * S + L; a.cs.width[2] = A_; a.cs.wrow[2] = 2 * S + 2 * L;
    a.act = GNN_ACT_TANH; a.Y = dY; a.ldy = S; a.thr = 1e9f; a.pred_flag = pred ? flag : nullptr; a.stat_part = part;
    std::vector<float> mu(K, 0.f);
    if (centred) { for (auto &x : mu) x = 0.4f + 0.2f * nd(rng); a.in_mean = up(mu); a.stat_shift = a.in_mean; }
    if constexpr (W32 == 3) gnn::k_train_fwd_b6<SQ, GNN_ACT_TANH><<<64, 64 * gnn::TB_WAVES, gnn::train_fwd_b6_lds<SQ>()>>>(a);
    else gnn::k_train_fwd<SQ, SQ><<<64, 64 * gnn::TB_WAVES, gnn::train_fwd_lds<SQ, SQ>()>>>(a);
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)M * S), pt(64 * 2 * S);
    CK(hipMemcpy(got.data(), dY, got.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(pt.data(), part, pt.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, bias_sum = 0, abs_sum = 0; std::vector<double> s1(S, 0), s2(S, 0);
    for (int m = 0; m < M; ++m) for (int h = 0; h < S; ++h) {
        double z = bf[h];
        for (int j = 0; j < S; ++j) z += ((double)st[(size_t)m * S + j] - mu[j]) * Wf[(size_t)j * S + h] + ((double)ag[(size_t)m * S + j] - mu[wa + j]) * Wf[(size_t)(wa + j) * S + h];
        for (int j = 0; j < L; ++j) z += ((double)xc[(size_t)m * 32 + j] - mu[S + j]) * Wf[(size_t)(S + j) * S + h] + ((double)xc[(size_t)m * 32 + L + j] - mu[2 * S + L + j]) * Wf[(size_t)(2 * S + L + j) * S + h];
        for (int j = 0; j < A_; ++j) z += ((double)xc[(size_t)m * 32 + 2 * L + j] - mu[2 * S + 2 * L + j]) * Wf[(size_t)(2 * S + 2 * L + j) * S + h];
        const double y = tanh(z), ys = y - (centred ? mu[h] : 0.0);          // (statistics partials are sums of y - stat_shift)
        worst = fmax(worst, fabs(y - got[(size_t)m * S + h])); s1[h] += ys; s2[h] += ys * ys;
        bias_sum += ((double)got[(size_t)m * S + h] - y) * (y > 0 ? 1.0 : -1.0); abs_sum += fabs((double)got[(size_t)m * S + h] - y);
    }
    double ws = 0;
    for (int h = 0; h < S; ++h) { double a1 = 0, a2 = 0; for (int b = 0; b < 64; ++b) { a1 += pt[(size_t)b * 2 * S + h]; a2 += pt[(size_t)b * 2 * S + S + h]; }
        ws = fmax(ws, fmax(fabs(a1 - s1[h]) / M, fabs(a2 - s2[h]) / M)); }
    int fl = -1; CK(hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost));
    printf("   (mean error signed by the value's sign %.3e, mean |error| %.3e)\n", bias_sum / ((double)M * S), abs_sum / ((double)M * S));
    printf("fwd<%d>%s M=%d pred=%d centred=%d  max abs err %.3e  stats err %.3e  flag %d\n", SQ, W32 == 3 ? " bf16x6" : W32 == 2 ? " 32x32p" : W32 ? " 32x32" : "", M, pred, centred, worst, ws, fl);
}


template <int SQ, bool W32, bool B6 = false>
void check_wgrad(int M, bool centred = false) {
    const int S = 16 * SQ, L = 7, A_ = 3, Kc = 2 * L + A_, K = 2 * S + Kc, wa = S + L;
    std::mt19937 rng(3); std::normal_distribution<float> nd(0, 1);
    std::vector<float> G((size_t)M * S), Y((size_t)M * S), st((size_t)M * S), ag((size_t)M * S), xc((size_t)M * 32, 0.f);
    for (auto *v : {&G, &st, &ag}) for (auto &x : *v) x = nd(rng);
    for (auto &x : Y) x = tanhf(nd(rng));
    for (int m = 0; m < M; ++m) { for (int j = 0; j < Kc; ++j) xc[(size_t)m * 32 + j] = nd(rng); xc[(size_t)m * 32 + Kc] = 1.f; }
    gnn::TrainWgradArgs a; memset(&a, 0, sizeof(a));
    const int n_wg = 37;
    a.M = M; a.rows_per_wg = B6 ? ((M + n_wg - 1) / n_wg + 63) / 64 * 64 : ((M + n_wg - 1) / n_wg + 15) / 16 * 16;
    const int grid = (M + a.rows_per_wg - 1) / a.rows_per_wg;
    float *part; CK(hipMalloc(&part, (size_t)grid * (K * S + S) * 4)); CK(hipMemset(part, 0, (size_t)grid * (K * S + S) * 4));
    a.G = up(G); a.Y = up(Y); a.act = GNN_ACT_TANH; a.state = up(st); a.agg = up(ag); a.xc = up(xc);
    a.K = K; a.wrow_state = 0; a.wrow_agg = wa; a.Kc = Kc; a.cs.n = 3; a.cs.width[0] = L; a.cs.wrow[0] = S; a.cs.width[1] = L; a.cs.wrow[1] = 2 * S + L; a.cs.width[2] = A_; a.cs.wrow[2] = 2 * S + 2 * L;
    a.part = part;
    std::vector<float> mu(K + 1, 0.f);
    if (centred) { for (int k = 0; k < K; ++k) mu[k] = 0.5f + 0.1f * nd(rng); a.mean = up(mu); }
    if constexpr (B6) {
        CK(hipFuncSetAttribute((const void *)gnn::k_train_wgrad_b6<SQ / 2, GNN_ACT_TANH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_wgrad_b6_lds<SQ / 2, GNN_ACT_TANH>()));
        gnn::k_train_wgrad_b6<SQ / 2, GNN_ACT_TANH><<<grid, 256, gnn::train_wgrad_b6_lds<SQ / 2, GNN_ACT_TANH>()>>>(a);
    }
    else if constexpr (W32) gnn::k_train_wgrad32<SQ / 2, GNN_ACT_TANH><<<grid, 256>>>(a); else gnn::k_train_wgrad<SQ><<<grid, 256>>>(a);
    CK(hipDeviceSynchronize());
    std::vector<float> pt((size_t)grid * (K * S + S));
    CK(hipMemcpy(pt.data(), part, pt.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> P((size_t)(K + 1) * S, 0.0);
    for (int m = 0; m < M; ++m) for (int h = 0; h < S; ++h) {
        const double y = Y[(size_t)m * S + h], dz = G[(size_t)m * S + h] * (1.0 - y * y);
        for (int j = 0; j < S; ++j) { P[(size_t)j * S + h] += (st[(size_t)m * S + j] - mu[j]) * dz; P[(size_t)(wa + j) * S + h] += (ag[(size_t)m * S + j] - mu[wa + j]) * dz; }
        for (int j = 0; j < L; ++j) { P[(size_t)(S + j) * S + h] += (xc[(size_t)m * 32 + j] - mu[S + j]) * dz; P[(size_t)(2 * S + L + j) * S + h] += (xc[(size_t)m * 32 + L + j] - mu[2 * S + L + j]) * dz; }
        for (int j = 0; j < A_; ++j) P[(size_t)(2 * S + 2 * L + j) * S + h] += (xc[(size_t)m * 32 + 2 * L + j] - mu[2 * S + 2 * L + j]) * dz;
        P[(size_t)K * S + h] += dz;
    }
    double worst = 0, scale = 0;
    for (size_t idx = 0; idx < P.size(); ++idx) { double sum = 0; for (int b = 0; b < grid; ++b) sum += pt[(size_t)b * (K * S + S) + idx]; worst = fmax(worst, fabs(sum - P[idx])); scale = fmax(scale, fabs(P[idx])); }
    printf("wgrad<%d>%s M=%d centred=%d  max abs err %.3e (scale %.3e)\n", SQ, B6 ? " bf16x6" : W32 ? " 32x32" : "", M, centred, worst, scale);
}

// the one-pass kernel against the two kernels it replaces (both checked against CPU loops above): same bits expected
template <int NB>
void check_fused(int M, bool bn, bool scale, int n_wg = 37) {
    const int S = 32 * NB, L = 7, A_ = 3, Kc = 2 * L + A_, K = 2 * S + Kc, wagg = S + L;
    std::mt19937 rng(11); std::normal_distribution<float> nd(0, 1);
    std::vector<float> G((size_t)M * S), st((size_t)M * S), ag((size_t)M * S), xc((size_t)M * 32, 0.f), W((size_t)K * S), gm(K), mu(K + 1, 0.f), va(K), m1(K), m2(K), rs(M);
    for (auto *v : {&G, &st, &ag, &W, &gm, &m1, &m2}) for (auto &x : *v) x = nd(rng);
    for (int k = 0; k < K; ++k) mu[k] = 0.5f + 0.1f * nd(rng);
    for (auto &x : va) x = fabsf(nd(rng)) + 0.1f;
    for (auto &x : rs) x = 0.5f + fabsf(nd(rng));
    for (int m = 0; m < M; ++m) { for (int j = 0; j < Kc; ++j) xc[(size_t)m * 32 + j] = nd(rng); xc[(size_t)m * 32 + Kc] = 1.f; }
    gnn::TrainWgradArgs a; memset(&a, 0, sizeof(a));
    a.M = M; a.rows_per_wg = ((M + n_wg - 1) / n_wg + 63) / 64 * 64;
    const int grid = (M + a.rows_per_wg - 1) / a.rows_per_wg;
    const size_t np = (size_t)grid * (K * S + S);
    float *part0, *part1; CK(hipMalloc(&part0, np * 4)); CK(hipMalloc(&part1, np * 4)); CK(hipMemset(part0, 0, np * 4)); CK(hipMemset(part1, 0, np * 4));
    a.G = up(G); a.Y = nullptr; a.act = GNN_ACT_LINEAR; a.state = up(st); a.agg = up(ag); a.xc = up(xc);
    a.K = K; a.wrow_state = 0; a.wrow_agg = wagg; a.Kc = Kc; a.cs.n = 3; a.cs.width[0] = L; a.cs.wrow[0] = S; a.cs.width[1] = L; a.cs.wrow[1] = 2 * S + L; a.cs.width[2] = A_; a.cs.wrow[2] = 2 * S + 2 * L;
    if (bn) a.mean = up(mu);
    gnn::TrainBwdArgs b; memset(&b, 0, sizeof(b));
    float *dx0, *dx1; CK(hipMalloc(&dx0, (size_t)M * 2 * S * 4)); CK(hipMalloc(&dx1, (size_t)M * 2 * S * 4)); CK(hipMemset(dx0, 0xff, (size_t)M * 2 * S * 4)); CK(hipMemset(dx1, 0xff, (size_t)M * 2 * S * 4));
    b.M = M; b.dZ = a.G; b.ldz = S; b.W = up(W); b.ldw = S; b.H = S; b.S = S; b.wrow_state = 0; b.wrow_agg = wagg; b.state = a.state; b.ld_state = S; b.agg = a.agg; b.ld_agg = S;
    if (bn) { b.gamma = up(gm); b.mean = a.mean; b.var = up(va); b.m1 = up(m1); b.m2 = up(m2); b.eps = 1e-3f; b.defer_state_bn = 1; }
    if (scale) b.agg_row_scale = up(rs);
    b.ld_dx = 2 * S; b.act = GNN_ACT_LINEAR;
    a.part = part0; b.dx = dx0;
    CK(hipFuncSetAttribute((const void *)gnn::k_train_wgrad_b6<NB, GNN_ACT_LINEAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_wgrad_b6_lds<NB, GNN_ACT_LINEAR>()));
    gnn::k_train_wgrad_b6<NB, GNN_ACT_LINEAR><<<grid, 256, gnn::train_wgrad_b6_lds<NB, GNN_ACT_LINEAR>()>>>(a);
    CK(hipFuncSetAttribute((const void *)gnn::k_train_bwd_dx_b6<2 * NB, GNN_ACT_LINEAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_bwd_b6_lds<2 * NB>()));
    gnn::k_train_bwd_dx_b6<2 * NB, GNN_ACT_LINEAR><<<64, 256, gnn::train_bwd_b6_lds<2 * NB>()>>>(b);
    CK(hipDeviceSynchronize());
    a.part = part1; b.dx = dx1;
    CK(hipFuncSetAttribute((const void *)gnn::k_train_wgrad_dx_b6<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_wgrad_dx_b6_lds<NB>()));
    gnn::k_train_wgrad_dx_b6<NB><<<grid, 256, gnn::train_wgrad_dx_b6_lds<NB>()>>>(a, b);
    CK(hipDeviceSynchronize());
    std::vector<float> p0(np), p1(np), d0((size_t)M * 2 * S), d1((size_t)M * 2 * S);
    CK(hipMemcpy(p0.data(), part0, np * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), part1, np * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(d0.data(), dx0, d0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d1.data(), dx1, d1.size() * 4, hipMemcpyDeviceToHost));
    {   // both partial sums against a CPU loop (state rows of P only: enough to tell which kernel is off)
        std::vector<double> Pc((size_t)S * S, 0.0);
        for (int m = 0; m < M; ++m) for (int j = 0; j < S; ++j) { const double x = st[(size_t)m * S + j] - (bn ? mu[j] : 0.f); for (int h = 0; h < S; ++h) Pc[(size_t)j * S + h] += x * G[(size_t)m * S + h]; }
        double e0 = 0, e1 = 0;
        for (int j = 0; j < S; ++j) for (int h = 0; h < S; ++h) { double s0 = 0, s1 = 0; for (int b_ = 0; b_ < grid; ++b_) { s0 += p0[(size_t)b_ * (K * S + S) + (size_t)j * S + h]; s1 += p1[(size_t)b_ * (K * S + S) + (size_t)j * S + h]; }
            e0 = fmax(e0, fabs(s0 - Pc[(size_t)j * S + h])); e1 = fmax(e1, fabs(s1 - Pc[(size_t)j * S + h])); }
        printf("   against a CPU loop (state rows): k_train_wgrad_b6 %.3e, one-pass kernel %.3e\n", e0, e1);
        std::vector<double> Pk((size_t)L * S, 0.0);          // the first constants segment: weight rows S .. S + L - 1 = line columns 0 .. L - 1
        for (int m = 0; m < M; ++m) for (int j = 0; j < L; ++j) { const double x = xc[(size_t)m * 32 + j] - (bn ? mu[S + j] : 0.f); for (int h = 0; h < S; ++h) Pk[(size_t)j * S + h] += x * G[(size_t)m * S + h]; }
        e0 = e1 = 0;
        for (int j = 0; j < L; ++j) for (int h = 0; h < S; ++h) { double s0 = 0, s1 = 0; for (int b_ = 0; b_ < grid; ++b_) { s0 += p0[(size_t)b_ * (K * S + S) + (size_t)(S + j) * S + h]; s1 += p1[(size_t)b_ * (K * S + S) + (size_t)(S + j) * S + h]; }
            e0 = fmax(e0, fabs(s0 - Pk[(size_t)j * S + h])); e1 = fmax(e1, fabs(s1 - Pk[(size_t)j * S + h])); }
        printf("   against a CPU loop (constants rows): k_train_wgrad_b6 %.3e, one-pass kernel %.3e\n", e0, e1);
    }
    double ep = 0, ed = 0, sp = 0, sd = 0; size_t nbad = 0;
    size_t worst_i = 0; size_t ndiffp = 0;
    for (size_t i = 0; i < np; ++i) { const double e = fabs((double)p0[i] - p1[i]); if (e > 0) ++ndiffp; if (e > ep) { ep = e; worst_i = i; } sp = fmax(sp, fabs((double)p0[i])); }
    if (ep > 0) { const size_t per = (size_t)K * S + S; printf("   P differs in %zu entries; worst: workgroup %zu, weight row %zu (state 0..%d, agg %d..%d, q = %d), column %zu: %g vs %g\n", ndiffp, worst_i / per, (worst_i % per) / S, S - 1, wagg, wagg + S - 1, K, (worst_i % per) % S, p0[worst_i], p1[worst_i]);
      size_t cnt_rows[4] = {0, 0, 0, 0}; for (size_t i = 0; i < np; ++i) if (p0[i] != p1[i]) { const size_t r = (i % per) / S; ++cnt_rows[r < (size_t)S ? 0 : (r >= (size_t)wagg && r < (size_t)(wagg + S)) ? 1 : r == (size_t)K ? 3 : 2]; }
      printf("   differing entries by kind: state rows %zu, agg rows %zu, constants rows %zu, q %zu\n", cnt_rows[0], cnt_rows[1], cnt_rows[2], cnt_rows[3]); }
    for (size_t i = 0; i < d0.size(); ++i) { const double e = fabs((double)d0[i] - d1[i]); if (!(e <= 1e-30)) ++nbad; ed = fmax(ed, e == e ? e : 1e30); sd = fmax(sd, fabs((double)d0[i])); }
    printf("fused<%d> M=%d bn=%d scale=%d grid=%d  P: max |diff| %.3e (scale %.3e)  dx: max |diff| %.3e (scale %.3e), %zu elements differ\n", NB, M, bn, scale, grid, ep, sp, ed, sd, nbad);
}

int main(int argc, char **argv) {
    if (argc > 1) { for (int nw : {256, 625, 100}) for (int sc = 0; sc < 2; ++sc) { check_fused<1>(40000, false, sc, nw); check_fused<2>(40000, false, sc, nw); check_fused<1>(40000, true, sc, nw); } return 0; }
    for (int M : {1000, 40000}) { check_fwd<1>(M, true, true); check_fwd<2>(M, true, true); check_fwd<4>(M, true, true); check_fwd<1, 3>(M, true, true); check_fwd<2, 3>(M, true, true); check_fwd<4, 3>(M, true, true); }
    for (int M : {1000, 40000, 77}) { check_wgrad<2, false>(M); check_wgrad<4, false>(M); check_wgrad<2, true>(M); check_wgrad<4, true>(M);
        check_wgrad<1, false>(M, true); check_wgrad<4, false>(M, true); check_wgrad<2, true>(M, true); check_wgrad<4, true>(M, true);
        check_wgrad<2, true, true>(M); check_wgrad<4, true, true>(M); check_wgrad<2, true, true>(M, true); check_wgrad<4, true, true>(M, true); }
    for (int M : {1000, 40000, 77}) for (int pr = 0; pr < 2; ++pr) { check_fwd<1>(M, pr); check_fwd<2>(M, pr); check_fwd<4>(M, pr); check_fwd<1, 3>(M, pr); check_fwd<2, 3>(M, pr); check_fwd<4, 3>(M, pr); }
    for (int M : {1000, 40000}) for (int bn = 0; bn < 2; ++bn) for (int sc = 0; sc < 2; ++sc) {
        check_bwd<1, 2>(M, bn, sc); check_bwd<2, 4>(M, bn, sc); check_bwd<4, 8>(M, bn, sc);
        check_bwd<1, 2, true>(M, bn, sc); check_bwd<2, 4, true>(M, bn, sc); check_bwd<4, 8, true>(M, bn, sc);
    }
    for (int M : {1000, 40000, 77}) for (int bn = 0; bn < 2; ++bn) for (int sc = 0; sc < 2; ++sc) { check_fused<2>(M, bn, sc); check_fused<1>(M, bn, sc); }
    return 0;
}
