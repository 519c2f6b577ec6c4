"""Batch assembly cost of a MUTAG epoch: reshuffle + re-merge of all batches (on_epoch_end), then the first touch of every
batch by the model (operand upload / CSR build), next to the device time of the epoch's forwards and train steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import starter_nets
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Models.training import Adam
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
device = torch.device('cuda', 0)
graphs = load_graphs()
for g in graphs: g.setAggregation('average')
kw = {}
if len(sys.argv) > 1: kw['assemble'] = sys.argv[1]
t0 = time.perf_counter(); seq = MultiGraphSequencer(graphs, 'g', 'average', 32, shuffle=True, device=device, **kw); torch.cuda.synchronize()
print(f'construct sequencer ({len(seq)} batches): {time.perf_counter() - t0:.3f} s')
ns, no = starter_nets(32, device, 'g')
gnn = GNNgraphBased(ns, no, 32, 50, 0.01)
gnn.compile(optimizer=Adam(0.01), loss='categorical_crossentropy')
for ep in range(3):
    np.random.seed(ep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seq.on_epoch_end(); torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(len(seq)): gnn(seq[i][0])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for i in range(len(seq)): gnn(seq[i][0])
    torch.cuda.synchronize(); t3 = time.perf_counter()
    for i in range(len(seq)): gnn.train_step(seq[i], seed=0)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f'epoch {ep}: on_epoch_end {t1-t0:.3f} s | first forward pass over the batches {t2-t1:.3f} s | second {t3-t2:.3f} s | train steps {t4-t3:.3f} s')
