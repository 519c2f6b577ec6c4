"""First GPU shake-out: MUTAG batch parity (fused + unfused) vs the oracle, then a quick ER timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased, GNNnodeBased, GNNarcBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
from oracle.harness import oracle_loop, rel_err

print(torch.cuda.get_device_name(0))
gs = load_graphs(limit=64)
for d, thr in [(32, 0.0), (32, 0.01), (0, 0.0), (64, 0.0)]:
    for focus, cls in [('g', GNNgraphBased), ('n', GNNnodeBased), ('a', GNNarcBased)]:
        gl = gs[:32]
        if focus != 'g':
            from gnnkeras_amd import GraphObject
            rng = np.random.default_rng(5)
            def mk(g):
                n = (g.nodes if focus == 'n' else g.arcs).shape[0]
                om = rng.random(n) < 0.7
                return GraphObject(nodes=g.nodes, arcs=g.arcs, targets=rng.normal(size=(int(om.sum()), 2)), focus=focus,
                                   set_mask=rng.random(n) < 0.8, output_mask=om)
            gl = [mk(g) for g in gl]
        seq = MultiGraphSequencer(gl, focus, 'average', 32, shuffle=False)
        x, y, sw = seq[0]
        inp, lay = get_inout_dims('state', 14, 3, 2, focus, d)
        ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
        inp, lay = get_inout_dims('output', 14, 3, 2, focus, d)
        no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
        gnn = cls(ns, no, d, 50 if d else 5, thr)
        N = x[0].shape[0]
        s0 = np.random.default_rng(1).normal(0, .1, (N, d)).astype(np.float32) if d else None
        k64, st64, o64 = oracle_loop(gnn, x, s0, np.float64)
        k32, st32, o32 = oracle_loop(gnn, x, s0, np.float32)
        for flags, nm in [(nat.FLAG_UNFUSED, 'unfused'), (0, 'fused')]:
            gnn.native_flags = flags
            k, st, o = gnn.Loop(*gnn.process_inputs(x), state0=None if s0 is None else torch.from_numpy(s0).cuda())
            torch.cuda.synchronize()
            print(f'd={d} thr={thr} focus={focus} {nm}: k={float(k)} (oracle {k32}/{k64}) state rel vs f32 {rel_err(st.cpu().numpy(), st32):.2e} '
                  f'vs f64 {rel_err(st.cpu().numpy(), st64):.2e} | out rel vs f32 {rel_err(o.cpu().numpy(), o32):.2e} vs f64 {rel_err(o.cpu().numpy(), o64):.2e} '
                  f'| oracle f32 vs f64 {rel_err(st32, st64):.2e}')
