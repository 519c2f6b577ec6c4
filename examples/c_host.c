/* A plain-C host of libgnnloop.so: no Python, no torch — the drop-in boundary of include/gnnloop.h used the way a
 * foreign runtime would use it (device pointers + sizes + a HIP stream).
 *
 * It builds a small directed graph (a ring with chords), runs the reference's node-focused Loop
 * (GNN/Models/GNN.py:245-274: state transition until convergence, then the output network) through
 * gnn_loop_forward, and checks (k, state, out) against a scalar double-precision restatement of the same recurrence
 * written right here (the reference's op sequence for a one-layer tanh state net and a linear output net).
 *
 * Build (tests/test_gpu_c_host.py does exactly this): a C99 compiler, the HIP runtime API header and library — no hipcc,
 * this file contains no device code:
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host.c -Lgnnkeras_amd/csrc -lgnnloop \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gnnkeras_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm -o c_host
 * Exit code 0 and a line "c_host: OK ..." on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gnnloop.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

enum { N = 200, L = 3, A = 2, D = 32, T = 2, MAX_IT = 80 };

static unsigned lcg_state = 12345u;
static float rnd(void) {                       /* uniform in (-1, 1), reproducible */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)((lcg_state >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

static void *upload(const void *src, size_t bytes) {
    void *d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
    if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void) {
    if (gnn_abi_version() != GNN_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }

    /* ---- graph: arcs (src -> dst) sorted by (dst, src): i-1 -> i, i-7 -> i, and i+3 -> i for even i ---------------- */
    static int asrc[3 * N], adst[3 * N], rowptr[N + 1];
    int E = 0;
    for (int j = 0; j < N; ++j) {
        int cand[3] = {(j + N - 1) % N, (j + N - 7) % N, (j % 2 == 0) ? (j + 3) % N : -1};
        rowptr[j] = E;
        for (int a = 0; a < N; ++a)            /* ascending source inside a destination */
            for (int c = 0; c < 3; ++c)
                if (cand[c] == a) { asrc[E] = a; adst[E] = j; ++E; }
    }
    rowptr[N] = E;
    static int arc_ids[3 * N];
    for (int e = 0; e < E; ++e) arc_ids[e] = e;  /* ArcNode: arc e feeds its destination; arcs are in (dst, src) order */

    static float nodes[N * L], arcs[3 * N * (2 + A)], state0[N * D];
    for (int i = 0; i < N * L; ++i) nodes[i] = rnd();
    for (int e = 0; e < E; ++e) {
        arcs[e * (2 + A) + 0] = (float)asrc[e]; arcs[e * (2 + A) + 1] = (float)adst[e];
        for (int a = 0; a < A; ++a) arcs[e * (2 + A) + 2 + a] = rnd();
    }
    for (int i = 0; i < N * D; ++i) state0[i] = 0.1f * rnd();

    /* ---- networks: state net Dense(2D + 2L + A -> D, tanh), output net Dense(D + L -> T, linear) -------------------- */
    enum { KS = 2 * D + 2 * L + A, KO = D + L };
    static float Ws[KS * D], bs[D], Wo[KO * T], bo[T];
    for (int i = 0; i < KS * D; ++i) Ws[i] = 0.04f * rnd();      /* small enough for a contraction: the loop stops by itself */
    for (int i = 0; i < D; ++i) bs[i] = 0.1f * rnd();
    for (int i = 0; i < KO * T; ++i) Wo[i] = 0.3f * rnd();
    for (int i = 0; i < T; ++i) bo[i] = 0.1f * rnd();
    const float threshold = 1e-3f;

    /* ---- scalar restatement in double (GNN.py:254-258 setup, :217-236 iteration, :196-214 condition, :273 output) -- */
    static double st[N * D], st_old[N * D], agg[N * D], agg_nodes[N * L], agg_arcs[N * A], out_ref[N * T];
    memset(agg_nodes, 0, sizeof agg_nodes); memset(agg_arcs, 0, sizeof agg_arcs);
    for (int e = 0; e < E; ++e) {
        for (int l = 0; l < L; ++l) agg_nodes[adst[e] * L + l] += nodes[asrc[e] * L + l];
        for (int a = 0; a < A; ++a) agg_arcs[adst[e] * A + a] += arcs[e * (2 + A) + 2 + a];
    }
    for (int i = 0; i < N * D; ++i) { st[i] = state0[i]; st_old[i] = 1.0; }
    int k_ref = 0;
    for (;;) {
        int moving = 0;                           /* condition: any node with ||s - s_old|| > thr ||s_old||, and k < max */
        for (int j = 0; j < N && !moving; ++j) {
            double d2 = 0, n2 = 0;
            for (int f = 0; f < D; ++f) { double d = st[j * D + f] - st_old[j * D + f]; d2 += d * d; n2 += st_old[j * D + f] * st_old[j * D + f]; }
            if (sqrt(d2) > (double)threshold * sqrt(n2)) moving = 1;
        }
        if (!moving || k_ref >= MAX_IT) break;
        memset(agg, 0, sizeof agg);
        for (int e = 0; e < E; ++e) for (int f = 0; f < D; ++f) agg[adst[e] * D + f] += st[asrc[e] * D + f];
        memcpy(st_old, st, sizeof st);
        for (int j = 0; j < N; ++j)
            for (int h = 0; h < D; ++h) {        /* inp = [state | nodes | agg_state | agg_nodes | agg_arcs] */
                double z = bs[h];
                int r = 0;
                for (int f = 0; f < D; ++f, ++r) z += st_old[j * D + f] * Ws[r * D + h];
                for (int l = 0; l < L; ++l, ++r) z += nodes[j * L + l] * Ws[r * D + h];
                for (int f = 0; f < D; ++f, ++r) z += agg[j * D + f] * Ws[r * D + h];
                for (int l = 0; l < L; ++l, ++r) z += agg_nodes[j * L + l] * Ws[r * D + h];
                for (int a = 0; a < A; ++a, ++r) z += agg_arcs[j * A + a] * Ws[r * D + h];
                st[j * D + h] = tanh(z);
            }
        ++k_ref;
    }
    for (int j = 0; j < N; ++j)
        for (int t = 0; t < T; ++t) {
            double z = bo[t];
            for (int f = 0; f < D; ++f) z += st[j * D + f] * Wo[f * T + t];
            for (int l = 0; l < L; ++l) z += nodes[j * L + l] * Wo[(D + l) * T + t];
            out_ref[j * T + t] = z;
        }

    /* ---- the same through the C ABI ----------------------------------------------------------------------------------- */
    static int out_index[N];
    for (int j = 0; j < N; ++j) out_index[j] = j;
    gnn_loop_args_t a;
    memset(&a, 0, sizeof a);
    a.abi_version = GNN_ABI_VERSION;
    a.n_nodes = N; a.n_arcs = E; a.dim_node_label = L; a.dim_arc_label = A;
    float *d_arcs = (float *)upload(arcs, sizeof(float) * (size_t)E * (2 + A));
    a.nodes = (const float *)upload(nodes, sizeof nodes); a.ld_nodes = L;
    a.arc_labels = d_arcs + 2; a.ld_arcs = 2 + A;
    a.adjacency.n_dst = N; a.adjacency.n_src = N; a.adjacency.nnz = E;
    a.adjacency.rowptr = (const int32_t *)upload(rowptr, sizeof rowptr);
    a.adjacency.src = (const int32_t *)upload(asrc, sizeof(int) * (size_t)E);
    a.arcnode.n_dst = N; a.arcnode.n_src = E; a.arcnode.nnz = E;
    a.arcnode.rowptr = a.adjacency.rowptr;
    a.arcnode.src = (const int32_t *)upload(arc_ids, sizeof(int) * (size_t)E);
    a.n_types = 1;
    a.net_state[0].in_dim = KS; a.net_state[0].n_layers = 1; a.net_state[0].units[0] = D;
    a.net_state[0].activation[0] = GNN_ACT_TANH;
    a.net_state[0].kernel[0] = (const float *)upload(Ws, sizeof Ws); a.net_state[0].bias[0] = (const float *)upload(bs, sizeof bs);
    a.net_output.in_dim = KO; a.net_output.n_layers = 1; a.net_output.units[0] = T;
    a.net_output.activation[0] = GNN_ACT_LINEAR;
    a.net_output.kernel[0] = (const float *)upload(Wo, sizeof Wo); a.net_output.bias[0] = (const float *)upload(bo, sizeof bo);
    a.state_dim = D; a.max_iteration = MAX_IT; a.state_threshold = threshold;
    a.state0 = (const float *)upload(state0, sizeof state0);
    a.focus = GNN_FOCUS_NODE; a.n_out = N; a.out_index = (const int32_t *)upload(out_index, sizeof out_index);
    float *d_k = NULL, *d_state = NULL, *d_out = NULL;
    CK(hipMalloc((void **)&d_k, sizeof(float))); CK(hipMalloc((void **)&d_state, sizeof(float) * N * D)); CK(hipMalloc((void **)&d_out, sizeof(float) * N * T));
    a.k_out = d_k; a.state_out = d_state; a.out = d_out;
    hipStream_t stream; CK(hipStreamCreate(&stream));
    a.stream = stream;
    a.workspace_bytes = gnn_loop_workspace_bytes(&a);
    if (a.workspace_bytes == 0) { fprintf(stderr, "workspace sizing failed: %s\n", gnn_last_error()); return 2; }
    CK(hipMalloc(&a.workspace, a.workspace_bytes));

    int worst = 0;
    const int paths[4] = {0, GNN_FLAG_UNFUSED, GNN_FLAG_FUSED_GEN2, GNN_FLAG_FUSED_GEN5};
    for (int pi = 0; pi < 4; ++pi) {
        a.flags = paths[pi];
        if (gnn_loop_forward(&a) != 0) { fprintf(stderr, "gnn_loop_forward: %s\n", gnn_last_error()); return 1; }
        CK(hipStreamSynchronize(stream));
        static float h_state[N * D], h_out[N * T];
        float h_k = -1.0f;
        CK(hipMemcpy(&h_k, d_k, sizeof h_k, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_state, d_state, sizeof h_state, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
        double es = 0, eo = 0, ms = 0, mo = 0;
        for (int i = 0; i < N * D; ++i) { es = fmax(es, fabs(h_state[i] - st[i])); ms = fmax(ms, fabs(st[i])); }
        for (int i = 0; i < N * T; ++i) { eo = fmax(eo, fabs(h_out[i] - out_ref[i])); mo = fmax(mo, fabs(out_ref[i])); }
        const int ok = ((int)h_k == k_ref) && es <= 1e-5 * ms && eo <= 1e-5 * mo;
        printf("c_host: flags=0x%02x k=%d (expected %d) state err %.2e out err %.2e %s\n", paths[pi], (int)h_k, k_ref, es / ms, eo / mo, ok ? "ok" : "MISMATCH");
        if (!ok) worst = 1;
    }
    if (worst) return 1;
    printf("c_host: OK  N=%d E=%d d=%d k=%d, workspace %zu bytes\n", N, E, D, k_ref, a.workspace_bytes);
    return 0;
}
