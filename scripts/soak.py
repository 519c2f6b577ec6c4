"""Soak test for the in-launch hand-offs: repeated runs must be bitwise reproducible, and the persistent whole-loop kernel
(generation 5) must agree with the one-launch-per-iteration kernel it shares its tile arithmetic with (generation 2): same
iteration count, states within 1e-5 (the two paths compute the per-node constant C in different summation orders - one
set-up launch against the general MFMA dense - so they are no longer bit-equal).  A race in the LDS ring (generation 4)
or in a grid barrier (generations 5 and 6) would show up as a run-to-run difference."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnnkeras_amd import _native as nat
from gnnkeras_amd.load_MUTAG import load_graphs
from gnnkeras_amd.synth import er_graph
from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
from gnnkeras_amd.Models.GNN import GNNgraphBased, GNNnodeBased
from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
# ---- MUTAG batches: generation 5 vs generation 2, bitwise -------------------------------------------------------------
gs = load_graphs()
for d, bs in ((32, 32), (64, 48)):
    seq = MultiGraphSequencer(gs, 'g', 'average', bs, shuffle=False)
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    gnn = GNNgraphBased(ns, no, d, 30, 0.02)
    t0 = time.time(); n = 0
    for rep in range(reps):
        for b in range(len(seq)):
            x = seq[b][0]
            inputs = gnn.process_inputs(x)
            s0 = torch.randn((x[0].shape[0], d), device='cuda') * 0.1
            res = {}
            for flag in (nat.FLAG_FUSED_GEN2, nat.FLAG_FUSED_GEN5, -1):
                gnn.native_flags = nat.FLAG_FUSED_GEN5 if flag < 0 else flag
                k, st, o = gnn.Loop(*inputs, state0=s0)
                res[flag] = (float(k), st.clone(), o.clone())
            a, c, c2 = res[nat.FLAG_FUSED_GEN2], res[nat.FLAG_FUSED_GEN5], res[-1]
            if c[0] != c2[0] or not torch.equal(c[1], c2[1]) or not torch.equal(c[2], c2[2]):
                bad += 1
                print(f'NOT REPRODUCIBLE d={d} rep={rep} batch={b}: k {c[0]} vs {c2[0]}, state max diff {float((c[1] - c2[1]).abs().max()):.3e}')
            rel = float((a[1] - c[1]).abs().max() / a[1].abs().max())
            if a[0] != c[0] or rel > 1e-5:
                bad += 1
                print(f'MISMATCH d={d} rep={rep} batch={b}: k {a[0]} vs {c[0]}, state rel diff {rel:.3e}')
            n += 1
    print(f'MUTAG d={d} batch={bs}: {n} forward pairs, {bad} mismatches, {time.time() - t0:.1f} s')
# ---- convergence groups: 16 batches as independent loops of one launch vs the batches one by one, bitwise -------------------
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
d = 32
inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'tanh', 'lecun_normal', 'lecun_normal', rng=0)
inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
gnn = GNNgraphBased(ns, no, d, 30, 0.02)
gnn.native_flags = nat.FLAG_FUSED_GEN5          # the form that spreads a group over several CUs: bit-equal to the single calls
t0 = time.time(); n = 0
for rep in range(reps):
    for bs in gnn._group_plan(seq, torch.device('cuda', 0)):
        if len(bs) < 2: continue
        i0 = bs[0]
        assert bs == list(range(i0, i0 + len(bs)))
        x, begin = seq.merged_batches(bs)
        s0 = torch.randn((begin[-1], d), device='cuda') * 0.1
        k, st, o = gnn.Loop(*gnn.process_inputs(x), state0=s0, groups=begin)
        r0 = 0
        for j in range(len(bs)):
            xb = seq[i0 + j][0]
            kb, stb, ob = gnn.Loop(*gnn.process_inputs(xb), state0=s0[begin[j]:begin[j + 1]].contiguous())
            rows = ob.shape[0]
            if float(kb) != float(k[j]) or not torch.equal(stb, st[begin[j]:begin[j + 1]]) or not torch.equal(ob, o[r0:r0 + rows]):
                bad += 1
                print(f'GROUP MISMATCH rep={rep} batch={i0 + j}: k {float(kb)} vs {float(k[j])}')
            r0 += rows
            n += 1
print(f'convergence groups: {n} batches compared, {time.time() - t0:.1f} s')
# ---- ER graph: generation 4 repeated, bitwise reproducible; persistent kernel at its size limit ----------------------------
for N, E, flag, name, d in ((200000, 2000000, nat.FLAG_FUSED_GEN4, 'generation 4', 64), (16000, 160000, nat.FLAG_FUSED_GEN5, 'generation 5', 64),
                             (30000, 300000, nat.FLAG_FUSED_GEN6, 'generation 6', 64), (60000, 600000, 0, 'state width 200 (k_state_xwide)', 200)):
    g = er_graph(N, E, aggregation_mode='average'); seq = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False); x = seq[0][0]
    inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    gnn = GNNnodeBased(ns, no, d, 20, 0.0); gnn.native_flags = flag
    inputs = gnn.process_inputs(x)
    s0 = torch.randn((N, d), device='cuda') * 0.1
    ref = None; t0 = time.time()
    for rep in range(reps * 5):
        k, st, o = gnn.Loop(*inputs, state0=s0)
        if ref is None: ref = (float(k), st.clone(), o.clone())
        elif float(k) != ref[0] or not torch.equal(st, ref[1]) or not torch.equal(o, ref[2]):
            bad += 1
            print(f'{name} NOT REPRODUCIBLE at rep {rep}: max diff {float((st - ref[1]).abs().max()):.3e}')
    print(f'{name} N={N}: {reps * 5} runs, k={ref[0]}, {time.time() - t0:.1f} s')

# ---- round 3: the persistent training kernels (two grid barriers per iteration each way), every MUTAG batch, run-to-run bit equality --------
from gnnkeras_amd.Models.training import LoopTrainer, SGD
seq = MultiGraphSequencer(gs, 'g', 'average', 32, shuffle=False)
for d, it in ((32, 30), (0, 5)):
    inp, lay = get_inout_dims('state', 14, 3, 2, 'g', d); ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'g', d); no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    gnn = GNNgraphBased(ns, no, d, it, 0.0)
    gnn.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    w0 = [a.copy() for a in ns.get_weights() + no.get_weights()]
    t0 = time.time(); n = 0
    for b in range(0, len(seq), max(1, len(seq) // (4 * reps))):
        x, y, sw = seq[b]
        s0 = torch.randn((x[0].shape[0], d), device='cuda') * 0.1 if d else None
        ref = None
        for rep in range(3):
            tr = LoopTrainer(gnn)
            r = tr.train_step(x, y, sw, state0=s0, apply=False)
            got = [r['loss'].clone(), r['y_pred'].clone()] + [g.clone() for g in tr.gs.gradients() + tr.go.gradients()]
            ns.set_weights(w0[:len(ns.get_weights())]); no.set_weights(w0[len(ns.get_weights()):])      # (the BN moving statistics moved)
            if ref is None: ref = got
            elif not all(torch.equal(a, b_) for a, b_ in zip(ref, got)): bad += 1; print(f'train step d={d} batch {b}: run {rep} differs')
            n += 1
    print(f'persistent training kernels d={d}: {n} steps, {time.time() - t0:.1f} s')
print('soak:', 'OK' if bad == 0 else f'{bad} FAILURES')
sys.exit(1 if bad else 0)
