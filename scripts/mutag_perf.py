"""MUTAG (C2) latency probe: wall time per batch vs device time per batch (HIP events) for the 136 batches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import starter_nets
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.GNN import GNNgraphBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
device = torch.device('cuda', 0)
graphs = load_graphs()
seq = MultiGraphSequencer(graphs, 'g', 'average', 32, shuffle=False, device=device)
ns, no = starter_nets(32, device, 'g')
gnn = GNNgraphBased(ns, no, 32, 50, 0.01)
items = [seq[i][0] for i in range(len(seq))]
inputs = [gnn.process_inputs(x) for x in items]
rng = np.random.default_rng(1)
s0s = [torch.from_numpy(rng.normal(0, 0.1, (x[0].shape[0], 32)).astype(np.float32)).to(device) for x in items]
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
for e in ev: e.record()
gnn.loop_events = ev
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for inp, s0 in zip(inputs, s0s): gnn.Loop(*inp, state0=s0)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f'rep {rep}: host enqueue {1e3*t_host:.1f} ms, total {1e3*t_all:.1f} ms for 136 batches -> {1e3*t_all/136:.3f} ms/batch, {1e3*t_all/4337:.4f} ms/graph')
loop_ms, tot_ms = [], []
for inp, s0 in zip(inputs, s0s):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gnn.Loop(*inp, state0=s0); b.record(); torch.cuda.synchronize()
    loop_ms.append(ev[0].elapsed_time(ev[1])); tot_ms.append(a.elapsed_time(b))
print(f'device: loop (50 iterations) median {np.median(loop_ms)*1e3:.0f} us = {np.median(loop_ms)*20:.1f} us/iter; whole forward median {np.median(tot_ms)*1e3:.0f} us')
# host cost of one forward: python marshalling + op dispatch + launches, with the device idle (sync after each call)
t_host = []
for inp, s0 in zip(inputs, s0s):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gnn.Loop(*inp, state0=s0)
    t_host.append(time.perf_counter() - t0)
print(f'host time of Loop() per batch (device idle): median {np.median(t_host)*1e6:.0f} us')
width = gnn._round_width(seq, device)
run = lambda i: gnn.Loop(*inputs[i], state0=s0s[i])
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in gnn._batches_concurrently(len(inputs), run, device, width): pass
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f'{width} streams: {1e3*t/136:.3f} ms/batch, {1e3*t/4337:.5f} ms/graph')
