"""Training-step latency on MUTAG batches (starter config and d=32 config) on the device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
gs = load_graphs(limit=32 * 20)
for g in gs: g.setAggregation('average')
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
for d, it in [(0, 5), (32, 50)]:
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    from gnnkeras_amd.Models.training import LoopTrainer
    for native in (True, False):
        gnn = GNNgraphBased(ns.clone(), no.clone(), d, it, 0.01)
        gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy', metrics=['accuracy'])
        gnn._trainer = LoopTrainer(gnn); gnn._trainer.use_native_step = native
        for i in range(len(seq)): gnn.train_step(seq[i], seed=0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(len(seq)): r = gnn.train_step(seq[i], seed=0)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / len(seq)
        print(f'd={d} max_iter={it} {"gnn_train_step (in-library)" if native else "building blocks from Python"}: train_step {t*1e3:.2f} ms/batch (k={r["k"]})')
