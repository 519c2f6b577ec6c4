"""Training-step latency on MUTAG batches (starter config and d=32 config) vs the torch-autograd CPU restatement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from oracle import torch_train
from oracle.harness import _np, _triple
gs = load_graphs(limit=32 * 20)
for g in gs: g.setAggregation('average')
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
for d, it in [(0, 5), (32, 50)]:
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    gnn = GNNgraphBased(ns, no, d, it, 0.01)
    gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
    for i in range(len(seq)): gnn.train_step(seq[i], seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(len(seq)): r = gnn.train_step(seq[i], seed=0)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / len(seq)
    x, y, sw = seq[0]
    nodes, arcs, _, sm, om, adj, an, ng = x
    s0 = np.random.default_rng(0).normal(0, .1, (nodes.shape[0], d)).astype(np.float32) if d else None
    t1 = time.perf_counter()
    torch_train.train_step(_np(nodes), _np(arcs), _triple(adj), _triple(an), _triple(ng), np.ones(nodes.shape[0], bool), net_state=ns.spec(), net_output=no.spec(),
                           state_vect_dim=d, max_iteration=it, state_threshold=0.01, focus='g', state0=s0, y=_np(y), sample_weight=_np(sw), loss='categorical_crossentropy', dtype=torch.float32)
    tc = time.perf_counter() - t1
    print(f'd={d} max_iter={it}: device train_step {t*1e3:.2f} ms/batch (k={r["k"]}), torch-autograd CPU float32 {tc*1e3:.1f} ms/batch')
