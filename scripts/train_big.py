"""Train-step robustness / timing on a large node-focused ER graph (default 1e5 nodes, 1e6 arcs, d = 64, 10 iterations)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNnodeBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
E = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
bn = (int(sys.argv[5]) != 0) if len(sys.argv) > 5 else True          # 0: networks without BatchNormalization (the one-pass backward kernel)
g = er_graph(N, E, aggregation_mode='average')
seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)
inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0, batch_normalization=bn)
inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1, batch_normalization=bn)
gnn = GNNnodeBased(ns, no, d, iters, 0.0)
gnn.compile(optimizer=Adam(0.001), loss='categorical_crossentropy', metrics=['accuracy'])
losses = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = gnn.train_step(seq[0], seed=0)
    torch.cuda.synchronize(); losses.append(float(r['loss']))
    print(f'step {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms  loss {losses[-1]:.5f} k={r["k"]}')
assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
print(f'ok: N={N} E={E} d={d} iterations={iters} batch_normalization={bn}; peak device memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB')
