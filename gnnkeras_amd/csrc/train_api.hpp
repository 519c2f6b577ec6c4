// C-ABI wrappers of the training building blocks (kernels_train.hpp) + the forward pieces the training loop drives
// one iteration at a time.  Included at the end of gnnloop.hip (shares its helpers).  Declared in include/gnnloop.h.
#pragma once
#include "kernels_train.hpp"

namespace {
inline int rows_per_chunk_for(int M, int *n_chunks) {
    int rpc = 64;                                    // multiple of 64 (the row step of k_dense_grad_partial)
    while ((long)cdiv(M, rpc) > 1024) rpc *= 2;      // at most 1024 partials
    *n_chunks = std::max(1, cdiv(M, rpc));
    return rpc;
}
}  // namespace

extern "C" {

int gnn_dense(const gnn_dense_args_t *d) {
    if (!d) return fail("args is NULL");
    if (d->n_segments < 1 || d->n_segments > GNN_MAX_SEGMENTS) return fail("n_segments out of [1,%d]", GNN_MAX_SEGMENTS);
    if (d->M < 0 || d->H < 1) return fail("bad M / H");
    if (d->activation < 0 || d->activation > GNN_ACT_SOFTMAX) return fail("unknown activation");
    if (d->M > 0 && (!d->W || !d->Y)) return fail("W / Y is NULL");
    gnn::SegDenseArgs a;
    memset(&a, 0, sizeof(a));
    a.gate = d->gate; a.M = d->M; a.H = d->H; a.nseg = d->n_segments;
    for (int s = 0; s < d->n_segments; ++s) {
        if (d->M > 0 && !d->seg_ptr[s]) return fail("segment %d pointer is NULL", s);
        a.seg[s] = gnn::Seg{d->seg_ptr[s], d->seg_rowidx[s], d->seg_ld[s], d->seg_width[s], d->seg_wrow[s]};
    }
    a.W = d->W; a.ldw = d->ldw > 0 ? d->ldw : d->H; a.bias = d->bias;
    a.addend = d->addend; a.ld_add = d->ld_addend; a.add_rowidx = d->addend_rowidx;
    a.in_center = d->in_center;
    const bool thin_softmax = d->activation == GNN_ACT_SOFTMAX && thin_dense_applies(a);      // finished in the same launch
    a.act = (d->activation == GNN_ACT_SOFTMAX && !thin_softmax) ? GNN_ACT_LINEAR : d->activation;
    a.Y = d->Y; a.ldy = d->ldy; a.out_rowidx = d->out_rowidx;
    hipStream_t st = (hipStream_t)d->stream;
    TRY(launch_segdense(a, st));
    if (d->activation == GNN_ACT_SOFTMAX && !thin_softmax) TRY(launch_softmax(d->gate, d->Y, d->M, d->H, d->ldy, d->out_rowidx, st));
    return 0;
}

int gnn_fold_bn(const float *W, const float *b, int32_t K, int32_t H, const float *gamma, const float *beta,
                const float *mean, const float *var, float eps, float *Wf, float *bf, int32_t centred, void *stream) {
    if (!W || !Wf || !bf || K < 1 || H < 1) return fail("bad arguments");
    FoldList fl;
    gnn::FoldJob &j = fl.fa.job[fl.fa.n_jobs++];
    j.centred = centred ? 1 : 0;
    j.W = W; j.b = b; j.K = K; j.H = H; j.gamma = gamma; j.beta = beta; j.mean = mean; j.var = var; j.eps = eps;
    j.Wf = Wf; j.bf = bf; j.blk_begin = 0;
    fl.blocks = H;
    return launch_fold_list(fl, (hipStream_t)stream);
}

size_t gnn_dense_grad_workspace_bytes(int32_t K, int32_t H, int32_t M) {
    int n_chunks;
    rows_per_chunk_for(std::max(M, 1), &n_chunks);
    return ((size_t)n_chunks * ((size_t)K * H + H) * sizeof(float) + 255) & ~(size_t)255;
}

int gnn_dense_grad(const float *X, int32_t ldx, const int32_t *rowidx, int32_t K, const float *dZ, int32_t ldz, int32_t H,
                   int32_t M, float *P, float *q, int32_t accumulate, const float *center, void *workspace, size_t workspace_bytes,
                   void *stream) {
    if (K < 1 || H < 1 || M < 0 || !P) return fail("bad arguments");
    if (M > 0 && (!X || !dZ)) return fail("X / dZ is NULL");
    if (!workspace || workspace_bytes < gnn_dense_grad_workspace_bytes(K, H, M)) return fail("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int n_chunks;
    const int rpc = rows_per_chunk_for(std::max(M, 1), &n_chunks);
    float *part = (float *)workspace;                   // [chunk][K*H + H]
    if (M == 0) {
        if (!accumulate) { HIP_OK(hipMemsetAsync(P, 0, sizeof(float) * K * H, st)); if (q) HIP_OK(hipMemsetAsync(q, 0, sizeof(float) * H, st)); }
        return 0;
    }
    dim3 grid(n_chunks, cdiv(K, 64), cdiv(H, 64));
    gnn::k_dense_grad_partial<<<grid, 256, 0, st>>>(X, ldx, rowidx, K, dZ, ldz, H, M, rpc, part, q ? 1 : 0, center);
    LAUNCH_OK();
    const int n = K * H + H;                            // P and q leave in one reduction launch
    gnn::k_reduce_partials<<<cdiv(n, 64), 256, 0, st>>>(part, n_chunks, n, P, accumulate, 1.0f, K * H, q);
    LAUNCH_OK();
    return 0;
}

int gnn_act_grad(const float *G, int32_t ldg, const float *Y, int32_t ldy, float *dZ, int32_t ldz, int32_t M, int32_t H,
                 int32_t activation, void *stream) {
    if (M < 0 || H < 1 || activation < 0 || activation > GNN_ACT_SOFTMAX) return fail("bad arguments");
    if (M == 0) return 0;
    if (!G || !Y || !dZ) return fail("NULL pointer");
    const long total = activation == GNN_ACT_SOFTMAX ? M : (long)M * H;
    gnn::k_act_grad<<<std::min(cdiv(total, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(G, ldg, Y, ldy, dZ, ldz, M, H, activation);
    LAUNCH_OK();
    return 0;
}

size_t gnn_colstats_workspace_bytes(int32_t K, int32_t M) {
    int n_chunks;
    rows_per_chunk_for(std::max(M, 1), &n_chunks);
    return ((size_t)(n_chunks + 1) * K * sizeof(float) + 255) & ~(size_t)255;
}

int gnn_colstats(const float *X, int32_t ldx, const int32_t *rowidx, int32_t K, int32_t M, float *mean, float *var,
                 float *moving_mean, float *moving_var, float momentum, const int32_t *gate, void *workspace,
                 size_t workspace_bytes, void *stream) {
    if (K < 1 || M < 1 || !X || !mean || !var) return fail("bad arguments (batch statistics need at least one row)");
    if (!workspace || workspace_bytes < gnn_colstats_workspace_bytes(K, M)) return fail("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (M <= 8192) {                                 // small batches: everything in one launch
        gnn::k_colstats_small<<<K, 256, 0, st>>>(X, ldx, rowidx, K, M, mean, var, moving_mean, moving_var, momentum, gate);
        LAUNCH_OK();
        return 0;
    }
    int n_chunks;
    const int rpc = rows_per_chunk_for(M, &n_chunks);
    float *part = (float *)workspace;
    gnn::k_colstats_partial<<<n_chunks, 256, 0, st>>>(X, ldx, rowidx, K, M, rpc, nullptr, part);
    LAUNCH_OK();
    gnn::k_reduce_partials<<<cdiv(K, 64), 256, 0, st>>>(part, n_chunks, K, mean, 0, 1.0f / (float)M, K, nullptr);
    LAUNCH_OK();
    gnn::k_colstats_partial<<<n_chunks, 256, 0, st>>>(X, ldx, rowidx, K, M, rpc, mean, part);
    LAUNCH_OK();
    gnn::k_reduce_partials<<<cdiv(K, 64), 256, 0, st>>>(part, n_chunks, K, var, 0, 1.0f / (float)M, K, nullptr);
    LAUNCH_OK();
    if (moving_mean && moving_var) {
        gnn::k_bn_moving_update<<<cdiv(K, 256), 256, 0, st>>>(mean, var, K, moving_mean, moving_var, momentum, gate);
        LAUNCH_OK();
    }
    return 0;
}

int gnn_first_layer_param_grads(const float *P, const float *q, const float *W, int32_t K, int32_t H, const float *gamma,
                                const float *beta, const float *mean, const float *var, float eps, int32_t M, float *dW,
                                float *db, float *dgamma, float *dbeta, float *m1, float *m2, int32_t accumulate, int32_t centered,
                                void *stream) {
    if (!P || !q || !W || !dW || K < 1 || H < 1 || M < 1) return fail("bad arguments");
    if (gamma && (!beta || !mean || !var || !dgamma || !dbeta)) return fail("BatchNormalization arrays are NULL");
    gnn::k_first_layer_param_grads<<<K, 64, 0, (hipStream_t)stream>>>(
        P, q, W, K, H, gamma, beta, mean, var, eps, 1.0f / (float)M, dW, db, dgamma, dbeta, m1, m2, accumulate, 1, (gamma && centered) ? 1 : 0);
    LAUNCH_OK();
    return 0;
}

int gnn_bn_input_grad(const float *dy, int32_t ld_dy, const float *x, int32_t ld_x, const int32_t *x_rowidx, int32_t M, int32_t width, int32_t k0,
                      const float *gamma, const float *mean, const float *var, float eps, const float *m1, const float *m2,
                      float *dx, int32_t ld_dx, void *stream) {
    if (M < 0 || width < 1) return fail("bad arguments");
    if (M == 0) return 0;
    if (!dy || !dx || (gamma && (!x || !mean || !var || !m1 || !m2))) return fail("NULL pointer");
    gnn::k_bn_input_grad<<<std::min(cdiv((long)M * width, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(
        dy, ld_dy, x, ld_x, x_rowidx, M, width, k0, gamma, mean, var, eps, m1, m2, dx, ld_dx);
    LAUNCH_OK();
    return 0;
}

int gnn_scatter_add_rows(const float *D, int32_t ldd, const int32_t *idx, int32_t M, int32_t width, float *G, int32_t ldg, void *stream) {
    if (M < 0 || width < 1) return fail("bad arguments");
    if (M == 0) return 0;
    if (!D || !G) return fail("NULL pointer");
    gnn::k_scatter_add_rows<<<std::min(cdiv((long)M * width, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(D, ldd, idx, M, width, G, ldg);
    LAUNCH_OK();
    return 0;
}

int gnn_axpby(float a, const float *x, float b, const float *y, float *out, size_t n, void *stream) {
    if (n == 0) return 0;
    if (!x || !out) return fail("NULL pointer");
    gnn::k_axpby<<<std::min(cdiv((long)n, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(a, x, b, y, out, n);
    LAUNCH_OK();
    return 0;
}

int gnn_dropout(const float *x, int32_t ldx, float *y, int32_t ldy, int32_t M, int32_t H, float rate, uint32_t key, int32_t alpha,
                int32_t backward, void *stream) {
    if (M < 0 || H < 1 || ldx < H || ldy < H || !(rate >= 0.0f && rate < 1.0f)) return fail("bad arguments (0 <= rate < 1)");
    if (M == 0) return 0;
    if (!x || !y) return fail("NULL pointer");
    const double t = (double)rate * 4294967296.0;
    const unsigned thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    float mul, fill = 0.0f, add = 0.0f;
    if (alpha) {
        const double ap = -1.6732632423543772 * 1.0507009873554805;
        const double a = 1.0 / sqrt((1.0 - rate) * (1.0 + rate * ap * ap)), b = -a * ap * rate;
        mul = (float)a;
        if (!backward) { fill = (float)(a * ap + b); add = (float)b; }
    } else {
        mul = (float)(1.0 / (1.0 - (double)rate));
    }
    gnn::k_dropout<<<(int)std::min<long>(cdiv((long)M * H, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, M, H, key, thr, mul, fill, add);
    LAUNCH_OK();
    return 0;
}

int gnn_loss_grad(int32_t kind, const float *y, const float *p, const float *sample_weight, int32_t M, int32_t T, float *dp,
                  float *loss_rows, void *stream) {
    if (kind < 0 || kind > 3 || M < 0 || T < 1) return fail("bad arguments");
    if (M == 0) return 0;
    if (!y || !p || !dp || !loss_rows) return fail("NULL pointer");
    gnn::k_loss_grad<<<cdiv(M, 256), 256, 0, (hipStream_t)stream>>>(kind, y, p, sample_weight, M, T, dp, loss_rows);
    LAUNCH_OK();
    return 0;
}

int gnn_adam_step(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2, float eps,
                  int32_t step, const int32_t *gate, void *stream) {
    if (n == 0) return 0;
    if (!p || !g || !m || !v || step < 1) return fail("bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    gnn::k_adam<<<std::min(cdiv((long)n, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2, gate);
    LAUNCH_OK();
    return 0;
}

int gnn_adam_multi(float *const *p, const float *const *g, float *const *m, float *const *v, const size_t *n, int32_t n_vars, float lr,
                   float beta1, float beta2, float eps, int32_t step, const int32_t *gate, void *stream) {
    if (n_vars < 0 || (n_vars > 0 && (!p || !g || !m || !v || !n)) || step < 1) return fail("bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    for (int j0 = 0; j0 < n_vars; j0 += gnn::ADAM_MAX_JOBS) {
        gnn::AdamJobs a;
        memset(&a, 0, sizeof(a));
        a.blk_begin[0] = 0;
        for (int j = j0; j < std::min(n_vars, j0 + gnn::ADAM_MAX_JOBS); ++j) {
            if (n[j] == 0) continue;
            if (!p[j] || !g[j] || !m[j] || !v[j]) return fail("gnn_adam_multi: variable %d has a NULL array", j);
            if (n[j] >= ((size_t)1 << 32)) return fail("gnn_adam_multi: variable %d is too large", j);
            const int q = a.n_jobs++;
            a.p[q] = p[j]; a.g[q] = g[j]; a.m[q] = m[j]; a.v[q] = v[j]; a.n[q] = (unsigned)n[j];
            a.blk_begin[q + 1] = a.blk_begin[q] + (int)std::min<long>(cdiv((long)n[j], 256), 1024);
        }
        if (a.n_jobs == 0) continue;
        gnn::k_adam_multi<<<a.blk_begin[a.n_jobs], 256, 0, (hipStream_t)stream>>>(a, lr, beta1, beta2, eps, bc1, bc2, gate);
        LAUNCH_OK();
    }
    return 0;
}

int gnn_sgd_step(float *p, const float *g, float *velocity, size_t n, float lr, float momentum, const int32_t *gate, void *stream) {
    if (n == 0) return 0;
    if (!p || !g) return fail("bad arguments");
    gnn::k_sgd<<<std::min(cdiv((long)n, 256), 256 * 16), 256, 0, (hipStream_t)stream>>>(p, g, velocity, n, lr, momentum, gate);
    LAUNCH_OK();
    return 0;
}

/* predicate with an explicit gate and k bookkeeping (training loop, one iteration at a time): if *gate != 0 (or gate is
 * NULL): *flag |= any(...), *k_out = k_val.  flag must be zeroed by the caller. */
int gnn_converged_gated(const float *state, const float *state_old, int32_t n, int32_t dim, int32_t ld, float threshold,
                        const int32_t *gate, int32_t *flag, float *k_out, float k_val, void *stream) {
    if (n < 0 || dim < 1 || ld < dim || !flag) return fail("bad arguments");
    if (n > 0 && !state) return fail("state is NULL");
    return launch_converge(gate, state, state_old, n, dim, ld, ld, threshold, flag, k_out, k_val, (hipStream_t)stream);
}

int gnn_aggregate_gated(const gnn_csr_t *csr, const float *X, int32_t ldx, int32_t F, float *out, int32_t ldo,
                        const int32_t *gate, void *stream) {
    if (!csr) return fail("csr is NULL");
    if (F < 0 || ldx < F || ldo < F) return fail("bad F / leading dimensions");
    TRY(check_csr(*csr, "csr", csr->n_dst, csr->n_src));
    return launch_aggregate(gate, *csr, X, ldx, F, out, ldo, (hipStream_t)stream);
}

}  // extern "C"
