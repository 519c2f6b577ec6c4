// Batch assembly on the device (reference GNN/Sequencers/GraphSequencers.py:42-46, :123-127 `build_batches` / `on_epoch_end` ->
// `GraphObject.merge`, GNN/graph_class.py:386-413): a merged batch is the block-diagonal concatenation of its graphs - every
// array of the merged graph (labels, arcs, masks, targets, the by-destination CSRs of Adjacency / ArcNode / NodeGraph, their
// by-source forms for the backward pass) is a run of per-graph segments copied from the dataset's device-resident arrays with a
// per-segment offset added to the ids.  One launch does all of them: a descriptor per segment, workgroups dealt to descriptors
// by a prefix array.  HBM-bound byte moving (a MUTAG batch: ~0.25 MB); what matters is that it is ONE launch and one small
// host-to-device copy per batch instead of ~1.3 ms of numpy + uploads.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gnnloop.h"

namespace gnn {

constexpr int RC_CHUNK = 2048;        // elements per workgroup

__global__ void __launch_bounds__(256)
k_ragged_copy(const gnn_ragged_desc_t *__restrict__ desc, int n_desc, const int *__restrict__ blk_begin) {
    // which descriptor does this workgroup serve?  blk_begin[d] <= blockIdx.x < blk_begin[d + 1]
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk_begin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const gnn_ragged_desc_t d = desc[lo];
    const long base = (long)((int)blockIdx.x - blk_begin[lo]) * RC_CHUNK;
    const long end = base + RC_CHUNK < d.count ? base + RC_CHUNK : d.count;
    switch (d.kind) {
        case GNN_RC_COPY_F32: {
            const float *s = (const float *)d.src; float *o = (float *)d.dst;
            for (long i = base + threadIdx.x; i < end; i += 256) o[i] = s[i];
        } break;
        case GNN_RC_COPY_I32_ADD: {
            const int *s = (const int *)d.src; int *o = (int *)d.dst;
            for (long i = base + threadIdx.x; i < end; i += 256) o[i] = s[i] + d.iadd;
        } break;
        case GNN_RC_COPY_ROWS_ADD2: {          // float rows of `width` columns; columns 0 and 1 (arc ids, float like the reference) += fval
            const float *s = (const float *)d.src; float *o = (float *)d.dst;
            for (long i = base + threadIdx.x; i < end; i += 256) o[i] = (i % d.width) < 2 ? s[i] + d.fval : s[i];
        } break;
        case GNN_RC_FILL_F32: {
            float *o = (float *)d.dst;
            for (long i = base + threadIdx.x; i < end; i += 256) o[i] = d.fval;
        } break;
        case GNN_RC_FILL_I32: {
            int *o = (int *)d.dst;
            for (long i = base + threadIdx.x; i < end; i += 256) o[i] = d.iadd;
        } break;
        case GNN_RC_IOTA_I32: {
            int *o = (int *)d.dst;
            for (long i = base + threadIdx.x; i < end; i += 256) o[i] = d.iadd + (int)i;
        } break;
        case GNN_RC_COPY_U8: {
            const unsigned char *s = (const unsigned char *)d.src; unsigned char *o = (unsigned char *)d.dst;
            for (long i = base + threadIdx.x; i < end; i += 256) o[i] = s[i];
        } break;
        default: break;
    }
}

}  // namespace gnn
