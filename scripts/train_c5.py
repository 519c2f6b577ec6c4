"""Train-step timing of a heterogeneous model on a large graph (default BASELINE C5: 5e5 nodes, 5e6 arcs, 3 node types, d = 64, 10 iterations;
reference CompositeGNN.py:275-304) - what bench.py times as `training.c5_d64_k10`, stand-alone for rocprofv3.
usage: python scripts/train_c5.py [N E d iterations aggregation batch_normalization steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.synth import er_composite_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.CompositeGNN import CompositeGNNnodeBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import CompositeMultiGraphSequencer
arg = lambda i, default, conv=str: conv(sys.argv[i]) if len(sys.argv) > i else default
N, E = arg(1, 500000, lambda v: int(float(v))), arg(2, 5000000, lambda v: int(float(v)))
d, iters, mode, bn, steps = arg(3, 64, int), arg(4, 10, int), arg(5, 'average'), arg(6, 1, int) != 0, arg(7, 4, int)
dims = (14, 8, 4)
g = er_composite_graph(N, E, dim_node_label=dims, aggregation_mode=mode, seed=1234)
seq = CompositeMultiGraphSequencer([g], 'n', mode, 1, shuffle=False)
inp, lay = get_inout_dims('state', list(dims), 3, 2, 'n', d)
ns = [MLP(i, lay, 'selu', 'lecun_normal', 'lecun_normal', rng=t, batch_normalization=bn) for t, i in enumerate(inp)]
inp, lay = get_inout_dims('output', list(dims), 3, 2, 'n', d)
no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=9, batch_normalization=bn)
gnn = CompositeGNNnodeBased(ns, no, d, iters, 0.0)
gnn.compile(optimizer=Adam(0.001), loss='categorical_crossentropy', metrics=['accuracy'])
losses = []
for rep in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = gnn.train_step(seq[0], seed=0)
    torch.cuda.synchronize(); losses.append(float(r['loss']))
    print(f'step {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms  loss {losses[-1]:.5f} k={r["k"]}')
print('orchestration:', nat.lib().gnn_last_kernel_name().decode())
assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
print(f'ok: N={N} E={E} d={d} iterations={iters} aggregation={mode} batch_normalization={bn}; peak device memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB')
