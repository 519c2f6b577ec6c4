// Building blocks of the backward pass / train_step (reference GNN/Models/GNN.py:277-306: GradientTape through the
// unrolled loop, i.e. back-propagation through time over the k executed iterations).  All float32, deterministic
// (two-stage reductions, no float atomics except the documented row scatter-add).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"

namespace gnn {

// derivative of a Keras activation expressed with the layer OUTPUT y = act(x) (so pre-activations are never stored)
__device__ __forceinline__ float activate_grad_from_output(int act, float y) {
    switch (act) {
        case GNN_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case GNN_ACT_SELU: return y > 0.0f ? 1.0507009873554805f : y + 1.0507009873554805f * 1.6732632423543772f;
        case GNN_ACT_TANH: return 1.0f - y * y;
        case GNN_ACT_SIGMOID: return y * (1.0f - y);
        case GNN_ACT_ELU: return y > 0.0f ? 1.0f : y + 1.0f;
        case GNN_ACT_SOFTPLUS: return 1.0f - expf(-y);
        default: return 1.0f;
    }
}

// dZ[m, :] = G[m, :] (.) act'(Y[m, :]);  softmax rows: dZ = Y (.) (G - sum_j G_j Y_j).   In place allowed (dZ == G).
__global__ void __launch_bounds__(256)
k_act_grad(const float *__restrict__ G, int ldg, const float *__restrict__ Y, int ldy, float *__restrict__ dZ, int ldz,
           int M, int H, int act) {
    if (act == GNN_ACT_SOFTMAX) {
        const int m = blockIdx.x * blockDim.x + threadIdx.x;
        if (m >= M) return;
        float dot = 0.0f;
        for (int h = 0; h < H; ++h) dot = fmaf(G[(size_t)m * ldg + h], Y[(size_t)m * ldy + h], dot);
        for (int h = 0; h < H; ++h) dZ[(size_t)m * ldz + h] = Y[(size_t)m * ldy + h] * (G[(size_t)m * ldg + h] - dot);
        return;
    }
    const size_t total = (size_t)M * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / H;
        const int h = (int)(i % H);
        dZ[m * ldz + h] = G[m * ldg + h] * activate_grad_from_output(act, Y[m * ldy + h]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM on the f32 matrix cores:  P[k, h] = sum_m X[row(m), k] * dZ[m, h]   (K x H = X^T . dZ) and,
// from blocks with blockIdx.y == 0, q[h] = sum_m dZ[m, h].  Stage 1 writes one partial per row chunk, stage 2 sums
// the partials in a fixed order (bitwise reproducible).  grid = (row chunks, ceil(K/64), ceil(H/64)).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DG_LD = 80;          // == 16 (mod 32): conflict-free transposed A-fragment and B-fragment reads

__global__ void __launch_bounds__(256)
k_dense_grad_partial(const float *__restrict__ X, int ldx, const int *__restrict__ rowidx, int K,
                     const float *__restrict__ dZ, int ldz, int H, int M, int rows_per_chunk,
                     float *__restrict__ part, int want_q, const float *__restrict__ center = nullptr) {      // part: [chunk][K*H (P) + H (q)]
    __shared__ float Xs[64 * DG_LD];
    __shared__ float Zs[64 * DG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.y * 64, h0 = blockIdx.z * 64;
    const int m_beg = blockIdx.x * rows_per_chunk, m_end = min(M, m_beg + rows_per_chunk);
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float qacc = 0.0f;                                   // threads 0..63 own one column each

    for (int m0 = m_beg; m0 < m_end; m0 += 64) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int mm = i / 64, cc = i % 64, m = m0 + mm;
            float xv = 0.0f, zv = 0.0f;
            if (m < m_end) {
                if (k0 + cc < K) xv = X[(rowidx ? (size_t)rowidx[m] : (size_t)m) * ldx + k0 + cc] - (center ? center[k0 + cc] : 0.0f);
                if (h0 + cc < H) zv = dZ[(size_t)m * ldz + h0 + cc];
            }
            Xs[mm * DG_LD + cc] = xv;
            Zs[mm * DG_LD + cc] = zv;
        }
        __syncthreads();
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const float av = Xs[(4 * s4 + g) * DG_LD + 16 * wave + r];      // A[k = 16*wave + r][m = 4*s4 + g]
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Zs[(4 * s4 + g) * DG_LD + 16 * c + r], acc[c], 0, 0, 0);
        }
        if (want_q && blockIdx.y == 0 && tid < 64) {
            for (int mm = 0; mm < 64; ++mm) qacc += Zs[mm * DG_LD + tid];
        }
        __syncthreads();
    }
    float *Pp = part + (size_t)blockIdx.x * ((size_t)K * H + H);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int h = h0 + 16 * c + r;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int k = k0 + 16 * wave + 4 * g + reg;
            if (k < K && h < H) Pp[(size_t)k * H + h] = acc[c][reg];
        }
    }
    if (want_q && blockIdx.y == 0 && tid < 64 && h0 + tid < H) Pp[(size_t)K * H + h0 + tid] = qacc;
}

// out[i] (+)= scale * sum_c part[c][i]  (entries i >= n1 go to out2[i - n1]: P and q of a weight gradient in one launch).
// 64 outputs per workgroup; the chunks are dealt to 4 thread rows in contiguous
// quarters (each summed in chunk order, 8 loads in flight), and the quarters meet in LDS in order: a fixed summation
// tree, bitwise reproducible.
__device__ __forceinline__ void reduce_partials_body(const float *__restrict__ part, int n_chunks, int n, float *__restrict__ out, int accumulate, float scale,
                                                     int n1, float *__restrict__ out2) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    const int per = (n_chunks + 3) / 4, c_beg = ty * per, c_end = min(n_chunks, c_beg + per);
    float s = 0.0f;
    if (i < n) {
        int c = c_beg;
        for (; c + 8 <= c_end; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < c_end; ++c) s += part[(size_t)c * n + i];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        const float t = (((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx]) * scale;
        float *o = i < n1 ? out + i : out2 + (i - n1);
        if (i < n1 || out2) *o = accumulate ? *o + t : t;
    }
}
__global__ void __launch_bounds__(256)
k_reduce_partials(const float *__restrict__ part, int n_chunks, int n, float *__restrict__ out, int accumulate, float scale,
                  int n1, float *__restrict__ out2) {
    reduce_partials_body(part, n_chunks, n, out, accumulate, scale, n1, out2);
}
// ... of every node type's weight-gradient partials in one launch (heterogeneous models, train_composite_big.hpp): blockIdx.y = the type
struct ReduceT { const float *part; int n_chunks, n; float *out; int accumulate; float scale; int n1; float *out2; };
struct ReduceTypes { ReduceT t[GNN_MAX_TYPES]; };
__global__ void __launch_bounds__(256) k_reduce_partials_types(ReduceTypes m) {
    ReduceT a = m.t[0];
#pragma unroll
    for (int t = 1; t < GNN_MAX_TYPES; ++t) if ((int)blockIdx.y == t) a = m.t[t];
    if ((int)blockIdx.x * 64 >= a.n) return;
    reduce_partials_body(a.part, a.n_chunks, a.n, a.out, a.accumulate, a.scale, a.n1, a.out2);
}

// Column statistics of a SMALL matrix (a merged MUTAG batch: ~1 k rows) in one launch: one workgroup per column, rows
// strided over its 256 threads, both passes of tf.nn.moments (mean, then biased variance of the centred values) with
// a fixed-order LDS tree each, and the Keras moving-average update (momentum 0.99, gated) at the end.
__global__ void __launch_bounds__(256)
k_colstats_small(const float *__restrict__ X, int ldx, const int *__restrict__ rowidx, int K, int M,
                 float *__restrict__ mean, float *__restrict__ var, float *moving_mean, float *moving_var,
                 float momentum, const int *gate) {
    __shared__ float part[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    float xs[32];                                      // M <= 8192: at most 32 rows per thread, kept for the second pass
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int m = tid + 256 * i;
        xs[i] = m < M ? X[(rowidx ? (size_t)rowidx[m] : (size_t)m) * ldx + k] : 0.0f;
        s += xs[i];
    }
    part[tid] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    const float mu = part[0] / (float)M;
    __syncthreads();
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int m = tid + 256 * i;
        const float d = xs[i] - mu;
        if (m < M) s = fmaf(d, d, s);
    }
    part[tid] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const float v = part[0] / (float)M;
        mean[k] = mu; var[k] = v;
        if (moving_mean && moving_var && !gate_closed(gate)) {
            moving_mean[k] = moving_mean[k] * momentum + mu * (1.0f - momentum);
            moving_var[k] = moving_var[k] * momentum + v * (1.0f - momentum);
        }
    }
}

// column statistics of X[row(m), 0:K] over M rows, two passes like tf.nn.moments: pass 1 (center == NULL) partial sums
// of x, pass 2 partial sums of (x - mean)^2; one partial per row chunk, reduced in chunk order by k_reduce_partials.
__global__ void __launch_bounds__(256)
k_colstats_partial(const float *__restrict__ X, int ldx, const int *__restrict__ rowidx, int K, int M,
                   int rows_per_chunk, const float *__restrict__ center, float *__restrict__ part) {
    // 64 columns x 4 row groups per pass: a wave reads 256 contiguous bytes of one row; the 4 row groups meet in LDS in
    // group order (deterministic)
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int m_beg = blockIdx.x * rows_per_chunk, m_end = min(M, m_beg + rows_per_chunk);
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + tx;
        const float c = (center && k < K) ? center[k] : 0.0f;
        float s = 0.0f;
        if (k < K) {
            for (int m = m_beg + ty; m < m_end; m += 4) {
                const float x = X[(rowidx ? (size_t)rowidx[m] : (size_t)m) * ldx + k] - c;
                s = center ? fmaf(x, x, s) : s + x;
            }
        }
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && k < K) part[(size_t)blockIdx.x * K + k] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
        __syncthreads();
    }
}

// Keras moving-average update (momentum 0.99): moving = moving * momentum + batch * (1 - momentum)
// (tf.keras.layers.BatchNormalization, training=True), skipped when the gate word is 0.
__global__ void __launch_bounds__(256)
k_bn_moving_update(const float *__restrict__ mean, const float *__restrict__ var, int K, float *moving_mean,
                   float *moving_var, float momentum, const int *gate) {
    if (gate_closed(gate)) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    moving_mean[k] = moving_mean[k] * momentum + mean[k] * (1.0f - momentum);
    moving_var[k] = moving_var[k] * momentum + var[k] * (1.0f - momentum);
}

// Parameter gradients of [BatchNormalization +] first Dense from P = X^T dZ (K x H) and q = colsum(dZ) (H):
//   with y = a (.) x + c per column (a = gamma*rstd, c = beta - mean*a; a = 1, c = 0 without BN):
//   dW[k,h] (+)= a_k P[k,h] + c_k q[h]          db[h] (+)= q[h]
//   sum_r dy[r,k]       = (W q)_k               =: S1_k      -> dbeta_k (+)= S1_k
//   sum_r dy[r,k] x[r,k] = sum_h W[k,h] P[k,h]  =: S2_k      -> dgamma_k (+)= rstd_k (S2_k - mean_k S1_k)
//   m1_k = S1_k / M ,  m2_k = rstd_k (S2_k - mean_k S1_k) / M     (consumed by k_bn_input_grad)
// One 64-lane workgroup per input feature k (lanes stride over the H outputs, the two dot products meet in a fixed
// shuffle tree); workgroup 0 also writes the bias gradient.  grid = K.
// `centered` (with BatchNormalization only): P is already P - mean q^T = (X - mean)^T dZ - the large-graph weight-gradient kernels subtract
// the column mean from every row as it arrives, because a P - mean q formed HERE cancels the leading digits of two sums over 10^5 .. 10^6
// rows whenever the inputs have a mean of their own size (relu / sigmoid states): dW = a P_c + beta q, S2 - mean S1 = sum_h W P_c.
// `n_chunks` > 1: P and q are still chunk partials ([chunk][K*H + H], the output of k_dense_grad_partial*): each value is
// summed here in chunk order (small batches: saves the reduction launch); n_chunks <= 1: P [K x H] and q [H] are final.
__device__ __forceinline__ void first_layer_param_grads_body(const float *__restrict__ P, const float *__restrict__ q, const float *__restrict__ W, int K,
                          int H, const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                          float inv_m, float *__restrict__ dW, float *__restrict__ db, float *dgamma, float *dbeta,
                          float *m1, float *m2, int accumulate, int n_chunks, int centered) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const size_t cstride = (size_t)K * H + H;
    auto sum_chunks = [&](const float *base) -> float {         // 8 loads in flight, summed in chunk order
        float s = 0.0f;
        int c = 0;
        for (; c + 8 <= n_chunks; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(c + u) * cstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < n_chunks; ++c) s += base[(size_t)c * cstride];
        return s;
    };
    auto q_at = [&](int h) -> float { return sum_chunks(q + h); };
    auto p_at = [&](int h) -> float { return sum_chunks(P + (size_t)k * H + h); };
    if (n_chunks > 1) {
        if (k == 0 && db)
            for (int h = lane; h < H; h += 64) { const float qh = q_at(h); db[h] = accumulate ? db[h] + qh : qh; }
    } else if (k == 0 && db)
        for (int h = lane; h < H; h += 64) db[h] = accumulate ? db[h] + q[h] : q[h];
    if (k >= K) return;
    float a = 1.0f, c = 0.0f, rstd = 1.0f, mu = 0.0f;
    if (gamma) {
        rstd = 1.0f / sqrtf(var[k] + eps);
        mu = mean[k];
        a = gamma[k] * rstd;
        c = beta[k] - mu * a;
    }
    float S1 = 0.0f, S2 = 0.0f;
    for (int h = lane; h < H; h += 64) {
        const float w = W[(size_t)k * H + h];
        const float p = n_chunks > 1 ? p_at(h) : P[(size_t)k * H + h], qh = n_chunks > 1 ? q_at(h) : q[h];
        S1 = fmaf(w, qh, S1);
        S2 = fmaf(w, p, S2);
        const float gw = centered ? fmaf(a, p, beta[k] * qh) : a * p + c * qh;
        dW[(size_t)k * H + h] = accumulate ? dW[(size_t)k * H + h] + gw : gw;
    }
    if (gamma) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { S1 += __shfl_xor(S1, off, 64); S2 += __shfl_xor(S2, off, 64); }
        if (lane == 0) {
            const float dg = centered ? rstd * S2 : rstd * (S2 - mu * S1);
            dgamma[k] = accumulate ? dgamma[k] + dg : dg;
            dbeta[k] = accumulate ? dbeta[k] + S1 : S1;
            if (m1) { m1[k] = S1 * inv_m; m2[k] = dg * inv_m; }
        }
    }
}
__global__ void __launch_bounds__(64)
k_first_layer_param_grads(const float *__restrict__ P, const float *__restrict__ q, const float *__restrict__ W, int K,
                          int H, const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                          float inv_m, float *__restrict__ dW, float *__restrict__ db, float *dgamma, float *dbeta,
                          float *m1, float *m2, int accumulate, int n_chunks = 1, int centered = 0) {
    first_layer_param_grads_body(P, q, W, K, H, gamma, beta, mean, var, eps, inv_m, dW, db, dgamma, dbeta, m1, m2, accumulate, n_chunks, centered);
}
// ... of every node type's state network in one launch (heterogeneous models): blockIdx.y = the type, grid.x = the widest network's K
struct ParamGradsT {
    const float *P, *q, *W; int K, H; const float *gamma, *beta, *mean, *var; float eps, inv_m;
    float *dW, *db, *dgamma, *dbeta, *m1, *m2; int accumulate, n_chunks, centered;
};
struct ParamGradsTypes { ParamGradsT t[GNN_MAX_TYPES]; };
__global__ void __launch_bounds__(64) k_first_layer_param_grads_types(ParamGradsTypes m) {
    ParamGradsT a = m.t[0];
#pragma unroll
    for (int t = 1; t < GNN_MAX_TYPES; ++t) if ((int)blockIdx.y == t) a = m.t[t];
    if ((int)blockIdx.x >= a.K) return;       // (K = 0: a type without rows)
    first_layer_param_grads_body(a.P, a.q, a.W, a.K, a.H, a.gamma, a.beta, a.mean, a.var, a.eps, a.inv_m, a.dW, a.db, a.dgamma, a.dbeta, a.m1, a.m2,
                                 a.accumulate, a.n_chunks, a.centered);
}

// Input gradient through a training-mode BatchNormalization for the column block [k0, k0+width):
//   dx[r,j] = gamma rstd (dy[r,j] - m1 - xhat[r,j] m2),  xhat = (x - mean) rstd      (identity when gamma == NULL)
__global__ void __launch_bounds__(256)
k_bn_input_grad(const float *__restrict__ dy, int ld_dy, const float *__restrict__ x, int ld_x,
                const int *__restrict__ x_rowidx, int M, int width, int k0, const float *gamma, const float *mean, const float *var, float eps, const float *m1,
                const float *m2, float *__restrict__ dx, int ld_dx) {
    const size_t total = (size_t)M * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / width;
        const int j = (int)(i % width);
        float v = dy[m * ld_dy + j];
        if (gamma) {
            const int k = k0 + j;
            const float rstd = 1.0f / sqrtf(var[k] + eps);
            const float xhat = (x[(x_rowidx ? (size_t)x_rowidx[m] : m) * ld_x + j] - mean[k]) * rstd;
            v = gamma[k] * rstd * (v - m1[k] - xhat * m2[k]);
        }
        dx[m * ld_dx + j] = v;
    }
}

// G[idx[m], :width] += D[m, :width]   (float atomics: rows may repeat, e.g. arc endpoints; order-dependent rounding)
__global__ void __launch_bounds__(256)
k_scatter_add_rows(const float *__restrict__ D, int ldd, const int *__restrict__ idx, int M, int width,
                   float *__restrict__ G, int ldg) {
    const size_t total = (size_t)M * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / width;
        const int j = (int)(i % width);
        atomicAdd(&G[(size_t)(idx ? idx[m] : (int)m) * ldg + j], D[m * ldd + j]);
    }
}

// out = a * x + b * y  elementwise (accumulating gradients across iterations, scaling by 1/k)
__global__ void __launch_bounds__(256)
k_axpby(float a, const float *__restrict__ x, float b, const float *__restrict__ y, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a * x[i] + (y ? b * y[i] : 0.0f);
}

// ---- Dropout / AlphaDropout (Keras layers of the reference MLP builder, MLP.py:60-66) ---------------------------------------
// The mask is a pure function of (key, row, column): a counter-based hash, so the backward sweep regenerates the mask of any
// call from its key instead of storing it.  keep <=> hash >= rate * 2^32.
//   Dropout       forward  y = x * keep / (1 - rate)                      backward  dx = dy * keep / (1 - rate)
//   AlphaDropout  forward  y = a * (x * keep + alpha' * (1 - keep)) + b   backward  dx = dy * a * keep
//                 alpha' = -selu_alpha * selu_scale, a = ((1 - rate)(1 + rate alpha'^2))^-1/2, b = -a alpha' rate
__device__ __forceinline__ unsigned lowbias32(unsigned h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ bool dropout_keep(unsigned key, unsigned row, unsigned col, unsigned thr) {
    return lowbias32(lowbias32(key + row) ^ (col * 0x9E3779B1u)) >= thr;
}
__global__ void __launch_bounds__(256)
k_dropout(const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy, int M, int H, unsigned key, unsigned thr,
          float mul, float fill, float add) {      // y = keep ? x * mul + add : fill     (backward: add = fill = 0)
    const size_t total = (size_t)M * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned r = (unsigned)(i / H), c = (unsigned)(i % H);
        y[(size_t)r * ldy + c] = dropout_keep(key, r, c, thr) ? fmaf(x[(size_t)r * ldx + c], mul, add) : fill;
    }
}

// ---- losses (Keras semantics, reduction SUM_OVER_BATCH_SIZE with sample weights): value and d loss / d prediction ----
enum { LOSS_CCE = 0, LOSS_BCE = 1, LOSS_MSE = 2, LOSS_MAE = 3 };
__global__ void __launch_bounds__(256)
k_loss_grad(int kind, const float *__restrict__ y, const float *__restrict__ p, const float *__restrict__ sw, int M, int T,
            float *__restrict__ dp, float *__restrict__ loss_rows) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float eps = 1e-7f;
    const float wgt = (sw ? sw[m] : 1.0f) / (float)M;
    const float *yr = y + (size_t)m * T, *pr = p + (size_t)m * T;
    float *dr = dp + (size_t)m * T;
    float loss = 0.0f;
    if (kind == LOSS_CCE) {                       // p / sum(p), clip, -sum y log p
        float sum = 0.0f, ysum = 0.0f;
        for (int t = 0; t < T; ++t) { sum += pr[t]; ysum += yr[t]; }
        float corr = 0.0f;                         // sum_i y_i [unclipped_i] (derivative of the normalisation)
        for (int t = 0; t < T; ++t) {
            const float pn = pr[t] / sum;
            const bool inside = pn >= eps && pn <= 1.0f - eps;
            const float pc = fminf(fmaxf(pn, eps), 1.0f - eps);
            loss -= yr[t] * logf(pc);
            if (inside) corr += yr[t];
        }
        for (int t = 0; t < T; ++t) {
            const float pn = pr[t] / sum;
            const bool inside = pn >= eps && pn <= 1.0f - eps;
            dr[t] = wgt * ((inside ? -yr[t] / pr[t] : 0.0f) + corr / sum);
        }
        (void)ysum;
    } else if (kind == LOSS_BCE) {                // mean over the last axis
        for (int t = 0; t < T; ++t) {
            const float pc = fminf(fmaxf(pr[t], eps), 1.0f - eps);
            const bool inside = pr[t] >= eps && pr[t] <= 1.0f - eps;
            loss -= (yr[t] * logf(pc) + (1.0f - yr[t]) * logf(1.0f - pc)) / (float)T;
            dr[t] = inside ? wgt * (-(yr[t] / pc) + (1.0f - yr[t]) / (1.0f - pc)) / (float)T : 0.0f;
        }
    } else if (kind == LOSS_MSE) {
        for (int t = 0; t < T; ++t) { const float d = pr[t] - yr[t]; loss += d * d / (float)T; dr[t] = wgt * 2.0f * d / (float)T; }
    } else {
        for (int t = 0; t < T; ++t) { const float d = pr[t] - yr[t]; loss += fabsf(d) / (float)T; dr[t] = wgt * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / (float)T; }
    }
    loss_rows[m] = loss * (sw ? sw[m] : 1.0f);
}

// ---- optimizers (tf.keras.optimizers.Adam / SGD defaults; `step` counts from 1) ------------------------------------------
__global__ void __launch_bounds__(256)
k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, size_t n, float lr,
       float b1, float b2, float eps, float bc1, float bc2, const int *gate) {
    if (gate_closed(gate)) return;                 // (the gradients of a failed step: see gnn_train_args_t::grads_ok_dev)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = v[i] = b2 * v[i] + (1.0f - b2) * gi * gi;
        p[i] -= lr * sqrtf(bc2) / bc1 * mi / (sqrtf(vi) + eps);     // alpha_t = lr sqrt(1-b2^t)/(1-b1^t)
    }
}
// every variable of a model in ONE launch (a MUTAG-sized step spent 8 launches of ~5 us on its 8 variables): job j owns the workgroups
// [blk_begin[j], blk_begin[j + 1]); same arithmetic per element as k_adam
constexpr int ADAM_MAX_JOBS = 32;
struct AdamJobs { float *p[ADAM_MAX_JOBS]; const float *g[ADAM_MAX_JOBS]; float *m[ADAM_MAX_JOBS], *v[ADAM_MAX_JOBS]; unsigned n[ADAM_MAX_JOBS]; int blk_begin[ADAM_MAX_JOBS + 1]; int n_jobs; };
__global__ void __launch_bounds__(256)
k_adam_multi(AdamJobs a, float lr, float b1, float b2, float eps, float bc1, float bc2, const int *gate) {
    if (gate_closed(gate)) return;
    int j = 0;
    while (j + 1 < a.n_jobs && (int)blockIdx.x >= a.blk_begin[j + 1]) ++j;
    float *__restrict__ p = a.p[j]; const float *__restrict__ g = a.g[j]; float *__restrict__ m = a.m[j], *__restrict__ v = a.v[j];
    const unsigned n = a.n[j], nb = a.blk_begin[j + 1] - a.blk_begin[j];
    for (unsigned i = (blockIdx.x - a.blk_begin[j]) * 256u + threadIdx.x; i < n; i += nb * 256u) {
        const float gi = g[i];
        const float mi = m[i] = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = v[i] = b2 * v[i] + (1.0f - b2) * gi * gi;
        p[i] -= lr * sqrtf(bc2) / bc1 * mi / (sqrtf(vi) + eps);
    }
}
__global__ void __launch_bounds__(256)
k_sgd(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ mom, size_t n, float lr, float momentum, const int *gate) {
    if (gate_closed(gate)) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (mom) { const float vel = mom[i] = momentum * mom[i] - lr * g[i]; p[i] += vel; }
        else p[i] -= lr * g[i];
    }
}

}  // namespace gnn

// =====================================================================================================================
// Kernels of the in-library training step (train_loop.hpp: gnn_train_step): the same arithmetic as the building blocks
// above, with the per-segment launches of a layer folded into one launch each (a MUTAG-sized train step is bound by
// the NUMBER of launches: ~1 800 before, ~700 with these).
// =====================================================================================================================
namespace gnn {

// ---- column statistics of several segments of a virtual concatenation in ONE launch (M <= 8192 rows) -----------------
struct StatSegs {
    Seg seg[GNN_MAX_SEGS];            // wrow = position of the segment's first column in mean / var
    int n;
    int col_begin[GNN_MAX_SEGS + 1];  // workgroup ranges: segment s owns workgroups [col_begin[s], col_begin[s + 1])
};
__global__ void __launch_bounds__(256)
k_colstats_segs_small(const int *gate, StatSegs ss, int M, float *__restrict__ mean, float *__restrict__ var) {
    if (gate_closed(gate)) return;
    __shared__ float part[256];
    int s = 0;
#pragma unroll
    for (int t = 1; t < GNN_MAX_SEGS; ++t)
        if (t < ss.n && (int)blockIdx.x >= ss.col_begin[t]) s = t;
    Seg sg = ss.seg[0]; int cb = ss.col_begin[0];
#pragma unroll
    for (int t = 1; t < GNN_MAX_SEGS; ++t)
        if (s == t) { sg = ss.seg[t]; cb = ss.col_begin[t]; }
    const int k = blockIdx.x - cb, tid = threadIdx.x;
    float xs[32];
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int m = tid + 256 * i;
        xs[i] = m < M ? sg.ptr[(sg.rowidx ? (size_t)sg.rowidx[m] : (size_t)m) * sg.ld + k] : 0.0f;
        acc += xs[i];
    }
    part[tid] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    const float mu = part[0] / (float)M;
    __syncthreads();
    acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int m = tid + 256 * i;
        const float d = xs[i] - mu;
        if (m < M) acc = fmaf(d, d, acc);
    }
    part[tid] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0) { mean[sg.wrow + k] = mu; var[sg.wrow + k] = part[0] / (float)M; }
}

// ---- weight-gradient GEMM over a virtual concatenation: every segment's K-blocks in one launch -------------------------
struct GradSegs {
    Seg seg[GNN_MAX_SEGS];            // wrow = first row of the segment in P
    int n;
    int blk_begin[GNN_MAX_SEGS + 1];  // blockIdx.y ranges (64 rows of K per block)
    const float *mean;                // [K] by weight row or NULL: subtracted from X as it is staged - P arrives as (X - mean)^T dZ
                                      // (k_first_layer_param_grads(centered = 1): no P - mean q^T to cancel afterwards)
};
__global__ void __launch_bounds__(256)
k_dense_grad_partial_segs(GradSegs gs, int K, const float *__restrict__ dZ, int ldz, int H, int M, int rows_per_chunk,
                          float *__restrict__ part) {                 // part: [chunk][K*H (P) + H (q)]
    __shared__ float Xs[64 * DG_LD];
    __shared__ float Zs[64 * DG_LD];
    int s = 0;
#pragma unroll
    for (int t = 1; t < GNN_MAX_SEGS; ++t)
        if (t < gs.n && (int)blockIdx.y >= gs.blk_begin[t]) s = t;
    Seg sg = gs.seg[0]; int bb = gs.blk_begin[0];
#pragma unroll
    for (int t = 1; t < GNN_MAX_SEGS; ++t)
        if (s == t) { sg = gs.seg[t]; bb = gs.blk_begin[t]; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int k0 = ((int)blockIdx.y - bb) * 64, h0 = blockIdx.z * 64;
    const int m_beg = blockIdx.x * rows_per_chunk, m_end = min(M, m_beg + rows_per_chunk);
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float qacc = 0.0f;
    for (int m0 = m_beg; m0 < m_end; m0 += 64) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int mm = i / 64, cc = i % 64, m = m0 + mm;
            float xv = 0.0f, zv = 0.0f;
            if (m < m_end) {
                if (k0 + cc < sg.width) xv = sg.ptr[(sg.rowidx ? (size_t)sg.rowidx[m] : (size_t)m) * sg.ld + k0 + cc] - (gs.mean ? gs.mean[sg.wrow + k0 + cc] : 0.0f);
                if (h0 + cc < H) zv = dZ[(size_t)m * ldz + h0 + cc];
            }
            Xs[mm * DG_LD + cc] = xv;
            Zs[mm * DG_LD + cc] = zv;
        }
        __syncthreads();
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const float av = Xs[(4 * s4 + g) * DG_LD + 16 * wave + r];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Zs[(4 * s4 + g) * DG_LD + 16 * c + r], acc[c], 0, 0, 0);
        }
        if (blockIdx.y == 0 && tid < 64) {
            for (int mm = 0; mm < 64; ++mm) qacc += Zs[mm * DG_LD + tid];
        }
        __syncthreads();
    }
    float *Pp = part + (size_t)blockIdx.x * ((size_t)K * H + H);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int h = h0 + 16 * c + r;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int k = k0 + 16 * wave + 4 * g + reg;
            if (k < sg.width && h < H) Pp[(size_t)(sg.wrow + k) * H + h] = acc[c][reg];
        }
    }
    if (blockIdx.y == 0 && tid < 64 && h0 + tid < H) Pp[(size_t)K * H + h0 + tid] = qacc;
}

// The same product with ONE workgroup per row chunk covering every column of the virtual concatenation (K <= 192, H <= 64):
// the dZ tile is read once instead of once per 64-column block of every segment (five times for the starter layout
// [state | labels | agg | agg labels | agg arcs] - three of those blocks are 14, 14 and 3 columns wide), and a B fragment
// read from LDS feeds up to three MFMAs.  1 M rows, K = 160, H = 64: 922 us -> 470 us (profiles/r02_train_c4_kernel_stats*.csv); the next tile is fetched while this one multiplies.
constexpr int DGA_TM = 32, DGA_LDX = 208;      // 208 == 16 (mod 32): conflict-free transposed A-fragment reads for K <= 192
__global__ void __launch_bounds__(256)
k_dense_grad_allk(GradSegs gs, int K, const float *__restrict__ dZ, int ldz, int H, int M, int rows_per_chunk,
                  float *__restrict__ part) {                         // part: [chunk][K*H (P) + H (q)]
    // (Round 3 tried forming dZ = G (.) act'(Y) in this kernel's fetch and storing it back in place to drop the separate
    // activation-gradient pass: the store in the middle of the pipelined fetch serialises the loads behind it - a C4-size step
    // went 24.1 -> 29.1 ms, and 28.1 ms with the feature compiled in but switched off.  Not adopted: profiles/r03_notes.txt.)
    __shared__ float Xs[DGA_TM * DGA_LDX];
    __shared__ float Zs[DGA_TM * DG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int m_beg = blockIdx.x * rows_per_chunk, m_end = min(M, m_beg + rows_per_chunk);
    const int n_blk = (K + 15) / 16;                                   // 16-row blocks of P; wave w owns blocks w, w + 4, w + 8
    f32x4 acc[3][4];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float qacc = 0.0f;
    // this thread's column of each 64-column block of the virtual concatenation: segment resolved once
    const float *xptr[3]; const int *xidx[3]; int xld[3]; bool xon[3]; float xmu[3];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
        const int kv = cb * 64 + lane;
        Seg sg = gs.seg[0]; int start = gs.seg[0].width, sbeg = 0;
#pragma unroll
        for (int t = 1; t < GNN_MAX_SEGS; ++t) {
            if (t < gs.n && kv >= start) { sg = gs.seg[t]; sbeg = start; }
            if (t < gs.n) start += gs.seg[t].width;
        }
        xon[cb] = kv < K;
        xptr[cb] = sg.ptr + (kv - sbeg); xidx[cb] = sg.rowidx; xld[cb] = sg.ld;
        xmu[cb] = (gs.mean && kv < K) ? gs.mean[sg.wrow + (kv - sbeg)] : 0.0f;
    }
    // software pipeline: the next tile's global loads are in flight while this tile runs through the matrix cores
    float xr[3][DGA_TM / 4], zr[DGA_TM / 4];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
#pragma unroll
            for (int pass = 0; pass < DGA_TM / 4; ++pass) {
                const int m = m0 + wave + 4 * pass;
                xr[cb][pass] = 0.0f;
                if (cb * 64 < K && xon[cb] && m < m_end) xr[cb][pass] = xptr[cb][(xidx[cb] ? (size_t)xidx[cb][m] : (size_t)m) * xld[cb]] - xmu[cb];
            }
        }
#pragma unroll
        for (int pass = 0; pass < DGA_TM / 4; ++pass) {
            const int m = m0 + wave + 4 * pass;
            zr[pass] = (m < m_end && lane < H) ? dZ[(size_t)m * ldz + lane] : 0.0f;
        }
    };
    if (m_beg < m_end) fetch(m_beg);
    for (int m0 = m_beg; m0 < m_end; m0 += DGA_TM) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            if (cb * 64 >= K) break;
#pragma unroll
            for (int pass = 0; pass < DGA_TM / 4; ++pass) Xs[(wave + 4 * pass) * DGA_LDX + cb * 64 + lane] = xr[cb][pass];
        }
#pragma unroll
        for (int pass = 0; pass < DGA_TM / 4; ++pass) Zs[(wave + 4 * pass) * DG_LD + lane] = zr[pass];
        __syncthreads();
        if (m0 + DGA_TM < m_end) fetch(m0 + DGA_TM);
#pragma unroll 2
        for (int s4 = 0; s4 < DGA_TM / 4; ++s4) {
            float bv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = Zs[(4 * s4 + g) * DG_LD + 16 * c + r];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int blk = wave + 4 * b;
                if (blk < n_blk) {
                    const float av = Xs[(4 * s4 + g) * DGA_LDX + 16 * blk + r];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[c], acc[b][c], 0, 0, 0);
                }
            }
        }
        if (tid < 64) {
            for (int mm = 0; mm < DGA_TM; ++mm) qacc += Zs[mm * DG_LD + tid];
        }
        __syncthreads();
    }
    float *Pp = part + (size_t)blockIdx.x * ((size_t)K * H + H);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int blk = wave + 4 * b;
        if (blk >= n_blk) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int h = 16 * c + r;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int kv = 16 * blk + 4 * g + reg;                // virtual column -> row of P through its segment's wrow
                int wrow = gs.seg[0].wrow + kv, start = gs.seg[0].width;
#pragma unroll
                for (int t = 1; t < GNN_MAX_SEGS; ++t) {
                    if (t < gs.n && kv >= start) wrow = gs.seg[t].wrow + kv - start;
                    if (t < gs.n) start += gs.seg[t].width;
                }
                if (kv < K && h < H) Pp[(size_t)wrow * H + h] = acc[b][c][reg];
            }
        }
    }
    if (tid < 64 && tid < H) Pp[(size_t)K * H + tid] = qacc;
}

// ---- input gradients through a training-mode BatchNormalization for up to 4 column blocks, in place ---------------------
struct BnGradReq { float *dy; int ld_dy; const float *x; int ld_x; const int *rowidx; int width, k0; };
struct BnGradArgs { BnGradReq r[4]; int n, M; const float *gamma, *mean, *var, *m1, *m2; float eps; };
__global__ void __launch_bounds__(256) k_bn_input_grad_segs(BnGradArgs a) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= a.n) break;
        const BnGradReq rq = a.r[q];
        const size_t total = (size_t)a.M * rq.width;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const size_t m = i / rq.width;
            const int j = (int)(i % rq.width), k = rq.k0 + j;
            const float rstd = 1.0f / sqrtf(a.var[k] + a.eps);
            const float xhat = (rq.x[(rq.rowidx ? (size_t)rq.rowidx[m] : m) * rq.ld_x + j] - a.mean[k]) * rstd;
            float *p = rq.dy + m * rq.ld_dy + j;
            *p = a.gamma[k] * rstd * (*p - a.m1[k] - xhat * a.m2[k]);
        }
    }
}

// ---- the k moving-average updates of a network applied k times in a row, in order (Keras: one per call) ----------------
__global__ void __launch_bounds__(256)
k_bn_moving_multi(const float *__restrict__ stats, int stride, int steps, int K, float *moving_mean, float *moving_var, float momentum,
                  const int *gate = nullptr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K || gate_closed(gate)) return;
    float mm = moving_mean[c], mv = moving_var[c];
    for (int t = 0; t < steps; ++t) {
        mm = mm * momentum + stats[(size_t)t * stride + c] * (1.0f - momentum);
        mv = mv * momentum + stats[(size_t)t * stride + K + c] * (1.0f - momentum);
    }
    moving_mean[c] = mm; moving_var[c] = mv;
}

// *ok = 1 when no launch of the step raised its error word (err[1]: the persistent backward kernel's expired grid barrier), else 0:
// the gate of the step's moving-average updates and of the optimizer (gnn_train_args_t::grads_ok_dev)
__global__ void k_grads_ok(const float *__restrict__ err, int *__restrict__ ok) { *ok = (err == nullptr || err[1] == 0.0f) ? 1 : 0; }

// *bad = 1 when idx[0:n] is not 0, 1, .. n - 1 (the large-graph thin-head kernels need output row m == node m)
__global__ void __launch_bounds__(256) k_not_identity(const int *__restrict__ idx, int n, float *bad) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (idx[i] != i) *bad = 1.0f;
}

// rows[t][0:n] = src[0:n] for t < reps (the iteration-invariant columns of every iteration's statistics vector)
__global__ void __launch_bounds__(256) k_replicate(const float *__restrict__ src, int n, int reps, float *__restrict__ rows) {
    const size_t total = (size_t)n * reps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) rows[i] = src[i % n];
}

// out[0] = scale * sum(in[0:n]): one workgroup, fixed summation tree
__global__ void __launch_bounds__(256) k_sum_scale(const float *__restrict__ in, int n, float scale, float *__restrict__ out) {
    __shared__ float part[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s += in[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] * scale;
}

// Wt[h][k] = W[k][h]
__global__ void __launch_bounds__(256) k_transpose(const float *__restrict__ W, int K, int H, float *__restrict__ Wt) {
    const int total = K * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) Wt[(size_t)(i % H) * K + i / H] = W[i];
}

// out[j, :F] = addend[j, :F] + row_scale[j] * sum_e w_e X[src_e, :F]: the transposed aggregate of the backward sweep with
// the own-state gradient folded in (G_state = dx_state + Adj . dx_agg)
template <int G>
__global__ void __launch_bounds__(256)
k_aggregate_add(int n_dst, const int *__restrict__ rowptr, const int *__restrict__ src, const float *__restrict__ w,
                const float *__restrict__ row_scale, const float *__restrict__ X, int ldx, int F, const float *__restrict__ addend,
                int ld_add, float *__restrict__ out, int ldo) {
    const int lane = threadIdx.x % G;
    const int groups = blockDim.x / G;
    for (int j = blockIdx.x * groups + threadIdx.x / G; j < n_dst; j += gridDim.x * groups) {
        const int beg = rowptr[j], end = rowptr[j + 1];
        const float scale = row_scale ? row_scale[j] : 1.0f;
        for (int f = lane; f < F; f += G) {
            float acc = 0.0f;
            for (int e = beg; e < end; ++e) acc = w ? fmaf(w[e], X[(size_t)src[e] * ldx + f], acc) : acc + X[(size_t)src[e] * ldx + f];
            out[(size_t)j * ldo + f] = addend[(size_t)j * ld_add + f] + acc * scale;
        }
    }
}

// 2-D strided add: out[m, :w] += in[m, :w]
__global__ void __launch_bounds__(256) k_add2d(float *__restrict__ out, int ldo, const float *__restrict__ in, int ldi, int M, int w) {
    const size_t total = (size_t)M * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        out[(i / w) * ldo + i % w] += in[(i / w) * ldi + i % w];
}

}  // namespace gnn
