#!/usr/bin/env python3
"""Error of the large-graph training kernels against float64 autograd AS A FUNCTION OF THE ROW COUNT: the weight
gradient contracts over all M rows in float32 MFMA accumulators (k_train_wgrad_b6: three-term bf16 splits of both operands since the second
pass of round 5; k_train_wgrad32: f32 inputs), the forward / dZ . W^T products are three-term bf16 splits (k_train_fwd_b6 / k_train_bwd_dx_b6) -
measured at 40 k, 160 k, 640 k and 1 M rows instead of extrapolated from 40 k.

    python scripts/dev/train_error_vs_rows.py [--rows 40000,160000,640000,1000000] [--iterations 3] [--out profiles/r05_train_error_vs_rows.txt]

Per row count: an Erdos-Renyi graph with 10 arcs per node, d = 64, the starter networks (BatchNormalization + Dense selu / softmax,
bench.py's training configuration), ONE train step through `gnn_train_step` on (i) the default kernels, (ii) the default kernels with the weight gradient on f32-input
MFMAs (GNN_TRAIN_WGRAD_B6=0), (iii) the float32-input MFMA kernels throughout (GNN_TRAIN_BF16X6=0 GNN_TRAIN_WGRAD32=0: read once per process,
hence child processes) and (iv) the building-block orchestration,
against oracle/torch_train.py in float64 (iterations checkpointed).  Errors are max-norm, relative to the tensor's own largest entry."""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def build(M, K):
    import numpy as np
    from gnnkeras_amd.synth import er_graph
    from gnnkeras_amd.Models.MLP import MLP, get_inout_dims
    from gnnkeras_amd.Models.GNN import GNNnodeBased
    from gnnkeras_amd.Sequencers.GraphSequencers import MultiGraphSequencer
    d = 64
    g = er_graph(M, 10 * M, aggregation_mode='average', seed=4321)
    x, y, sw = MultiGraphSequencer([g], 'n', 'average', 1, shuffle=False)[0]
    inp, lay = get_inout_dims('state', 14, 3, 2, 'n', d)
    ns = MLP(inp[0], lay, 'selu', 'lecun_normal', 'lecun_normal', rng=0)
    inp, lay = get_inout_dims('output', 14, 3, 2, 'n', d)
    no = MLP(inp[0], lay, 'softmax', 'glorot_normal', 'glorot_normal', rng=1)
    model = GNNnodeBased(ns, no, d, K, 0.0)
    s0 = np.random.default_rng(1).normal(0, 0.1, (M, d)).astype(np.float32)
    return model, x, y, sw, s0


def oracle(M, K, path):
    import numpy as np
    from test_gpu_training import oracle_step
    model, x, y, sw, s0 = build(M, K)
    t0 = time.time()
    w = oracle_step(model, x, y, sw, s0, 'categorical_crossentropy', checkpoint_iterations=True)
    np.savez(path, loss=w['loss'], y_pred=w['y_pred'], state=w['state'], kinks=w['kinks_state'][0], t=time.time() - t0,
             **{f'gs{i}': a for i, a in enumerate(w['grads_state'])}, **{f'go{i}': a for i, a in enumerate(w['grads_output'])})


def device(M, K, path, native):
    import numpy as np, torch
    from gnnkeras_amd.Models.training import LoopTrainer, SGD
    from oracle.harness import rel_err
    model, x, y, sw, s0 = build(M, K)
    w = np.load(path)
    model.compile(optimizer=SGD(0.0), loss='categorical_crossentropy')
    tr = LoopTrainer(model); tr.use_native_step = bool(native)
    res = tr.train_step(x, y, sw, state0=torch.from_numpy(s0).cuda(), apply=False)
    torch.cuda.synchronize()
    names = ['gamma', 'beta', 'kernel', 'bias']
    out = {'rows': M, 'loss': abs(float(res['loss']) - float(w['loss'])) / max(1.0, abs(float(w['loss']))),
           'y_pred': rel_err(res['y_pred'].cpu().numpy(), w['y_pred']), 'state': rel_err(res['state'].cpu().numpy(), w['state']),
           'kink_elements': int(w['kinks'].sum()), 'oracle_s': float(w['t'])}
    for net, gs, key in (('state', tr.gs.gradients(), 'gs'), ('output', tr.go.gradients(), 'go')):
        for i, g in enumerate(gs):
            r = w[f'{key}{i}']
            out[f'{net}.{names[i]}'] = float(np.max(np.abs(g.cpu().numpy() - r)) / max(float(np.max(np.abs(r))), 1e-30))
    print('RESULT ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', default='40000,160000,640000,1000000')
    ap.add_argument('--iterations', type=int, default=3)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r05_train_error_vs_rows.txt'))
    ap.add_argument('--oracle', type=int, default=0); ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--npz', default=''); ap.add_argument('--native', type=int, default=1)
    a = ap.parse_args()
    if a.oracle: oracle(a.oracle, a.iterations, a.npz); sys.exit(0)
    if a.device: device(a.device, a.iterations, a.npz, a.native); sys.exit(0)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    lines = [f'# gradient / prediction / state errors of one train step (d = 64, BatchNormalization, selu, {a.iterations} iterations, 10 arcs per node) against float64 autograd,',
             '# max-norm relative to the tensor\'s own largest entry; kink = pre-activations of the float64 oracle within 1e-6 of the selu kink', '']
    cols = ['loss', 'y_pred', 'state', 'state.gamma', 'state.beta', 'state.kernel', 'state.bias', 'output.gamma', 'output.beta', 'output.kernel', 'output.bias']
    lines.append(f"{'rows':>9} {'kernels':<28} {'kink':>5} " + ' '.join(f'{c:>13}' for c in cols))
    variants = (('default (bf16x6 + wgrad_b6)', {}, 1), ('bf16x6 + wgrad32 (round 5a)', {'GNN_TRAIN_WGRAD_B6': '0'}, 1),
                ('f32-input MFMA kernels', {'GNN_TRAIN_BF16X6': '0', 'GNN_TRAIN_WGRAD32': '0'}, 1), ('building blocks (general)', {}, 0))
    procs = {}
    rows = [int(float(v)) for v in a.rows.split(',')]
    for M in rows:        # the oracles (host, minutes at 1 M) all at once; the device runs follow as each lands
        npz = f'/tmp/train_err_oracle_{M}.npz'
        procs[M] = (npz, subprocess.Popen([sys.executable, __file__, '--oracle', str(M), '--iterations', str(a.iterations), '--npz', npz]))
    for M in rows:
        npz, pr = procs[M]
        if pr.wait() != 0: lines.append(f'{M:>9} oracle failed'); continue
        for name, env, native in variants:
            r = subprocess.run([sys.executable, __file__, '--device', str(M), '--iterations', str(a.iterations), '--npz', npz, '--native', str(native)],
                               capture_output=True, text=True, env=dict(os.environ, **env))
            got = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
            if not got: lines.append(f'{M:>9} {name:<28} FAILED: {r.stderr[-300:]}'); continue
            o = json.loads(got[-1][7:])
            lines.append(f"{M:>9} {name:<28} {o['kink_elements']:>5} " + ' '.join(f'{o[c]:>13.2e}' for c in cols))
        open(a.out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
