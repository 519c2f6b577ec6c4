"""ORACLE for train_step — TEST INFRASTRUCTURE ONLY (see gnn_oracle.py header; parity UNPINNED by the reference: no
TensorFlow, no reference tests).

Restates the reference's `GNNnodeBased.train_step` (GNN/Models/GNN.py:277-306) with torch *autograd* on the CPU in
float64: the eager `GradientTape` of the reference becomes `loss.backward()` through the same unrolled op sequence
(adjoint SpMM, concat, training-mode BatchNormalization on batch statistics, Dense, activation, predicate). The product
has no autograd; its hand-written backward kernels are checked against the gradients computed here.
Keras semantics restated: BN training = biased batch variance, moving = moving*0.99 + batch*0.01 on every call;
`categorical_crossentropy` on probabilities = normalise, clip to [1e-7, 1-1e-7], -sum y log p; loss reduction
SUM_OVER_BATCH_SIZE with sample weights; Adam alpha_t = lr*sqrt(1-b2^t)/(1-b1^t), epsilon 1e-7 outside the sqrt.
Dropout / AlphaDropout (MLP.py:60-66, Keras layer semantics) take their keep masks from `dropout_keep_mask` below - a
numpy restatement of the counter hash include/gnnloop.h documents for gnn_dropout - so that the oracle and the device
draw the SAME masks and the gradients are comparable; the arithmetic around the mask is autograd's.
"""
from __future__ import annotations

import numpy as np
import torch

from .torch_cpu import ACT

EPS_BN, MOMENTUM = 1e-3, 0.99
KINK_ACTS = ('relu', 'selu')      # activations whose derivative JUMPS at 0 (relu 0 -> 1, selu 1.7581 -> 1.0507; elu's is continuous there)
KINK_TOL = 1e-6                   # a float32 implementation may put a pre-activation this close to 0 on the other side


def _lowbias32(h):
    h = np.asarray(h, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(15)); h = (h * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
    return h ^ (h >> np.uint64(16))


def mix32(*values):
    """Key of one Dropout layer call from (step seed, network id, call number, layer index)."""
    h = np.uint64(0x9E3779B9)
    for v in values: h = _lowbias32(h ^ (np.uint64(int(v) & 0xFFFFFFFF)))
    return int(h)


def dropout_keep_mask(key, M, H, rate):
    """keep[r, c] <=> lowbias32(lowbias32(key + r) ^ c * 0x9E3779B1) >= rate * 2^32   (include/gnnloop.h, gnn_dropout)"""
    r = np.arange(M, dtype=np.uint64)[:, None]
    c = np.arange(H, dtype=np.uint64)[None, :]
    h = _lowbias32(_lowbias32((np.uint64(key) + r) & np.uint64(0xFFFFFFFF)) ^ ((c * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)))
    thr = min(int(float(np.float32(rate)) * 4294967296.0), 4294967295)
    return h >= np.uint64(thr)


class Net:
    def __init__(self, spec, weights, dtype=torch.float64, net_id=0, step_seed=0):
        self.drop = {}
        pos = [int(v) for v in (spec.get('dropout_pos') or [])]
        for i, (r, q) in enumerate(zip(spec.get('dropout_rate') or [], pos)):
            if float(r) > 0: self.drop.setdefault(q, []).append((float(r), i))
        self.alpha, self.net_id, self.step_seed, self.calls = bool(spec.get('alphadropout')), net_id, step_seed, 0
        self.bn = spec['batch_normalization']
        self.acts = spec['activations']
        w = [torch.tensor(np.asarray(a), dtype=dtype) for a in weights]
        pos = 0
        if self.bn:
            self.gamma, self.beta = w[0].requires_grad_(), w[1].requires_grad_()
            self.moving_mean, self.moving_var = w[2].clone(), w[3].clone()
            pos = 4
        self.W = [w[pos + 2 * i].requires_grad_() for i in range(len(self.acts))]
        self.b = [w[pos + 2 * i + 1].requires_grad_() for i in range(len(self.acts))]
        self.kreg = spec.get('kernel_regularizer') or [None] * len(self.acts)
        self.breg = spec.get('bias_regularizer') or [None] * len(self.acts)
        # per Dense layer and output unit: how many pre-activations of this step came within KINK_TOL of an activation kink.  Where a
        # float32 implementation lands on the other side, that element's act'(z) differs by the jump and the unit's kernel column /
        # bias gradient by |G| x jump - not an arithmetic error of the implementation; the tests widen the bar for exactly those units.
        self.kinks = [np.zeros(int(W_.shape[1]), dtype=np.int64) for W_ in self.W]
        self._recompute = False       # (a checkpointed re-run of a call: no second moving-average update, no second kink count)

    def penalty(self):
        """Keras `layer.losses` of the Dense layers: l1 * sum|w| + l2 * sum(w^2), each variable once (MLP.py:48-49)."""
        tot = 0.0
        for regs, ws in ((self.kreg, self.W), (self.breg, self.b)):
            for r, w in zip(regs, ws):
                if r is not None: tot = tot + r[0] * w.abs().sum() + r[1] * (w * w).sum()
        return tot

    def trainable(self):
        v = [self.gamma, self.beta] if self.bn else []
        for W, b in zip(self.W, self.b): v += [W, b]
        return v

    def _dropout(self, x, q, call):
        """Keras Dropout / AlphaDropout layers sitting at position q (in front of Dense q; q = #Dense: behind the last)."""
        for rate, index in self.drop.get(q, []):
            keep = torch.from_numpy(dropout_keep_mask(mix32(self.step_seed, self.net_id, call, index), x.shape[0], x.shape[1], rate))
            r = float(np.float32(rate))
            if self.alpha:
                ap = -1.6732632423543772 * 1.0507009873554805
                a = ((1 - r) * (1 + r * ap ** 2)) ** -0.5
                x = a * torch.where(keep, x, torch.full_like(x, ap)) + (-a * ap * r)
            else:
                x = torch.where(keep, x / (1 - r), torch.zeros_like(x))
        return x

    def __call__(self, x, training=True):
        call = self.calls
        if training: self.calls += 1
        if self.bn:
            if training:
                mean = x.mean(0)
                var = ((x - mean) ** 2).mean(0)
                if not self._recompute:
                    with torch.no_grad():
                        self.moving_mean.mul_(MOMENTUM).add_(mean * (1 - MOMENTUM))
                        self.moving_var.mul_(MOMENTUM).add_(var * (1 - MOMENTUM))
            else:
                mean, var = self.moving_mean, self.moving_var
            x = (x - mean) / torch.sqrt(var + EPS_BN) * self.gamma + self.beta
        for l, (W, b, a) in enumerate(zip(self.W, self.b, self.acts)):
            if training: x = self._dropout(x, l, call)
            z = x @ W + b
            if a in KINK_ACTS and not self._recompute:
                with torch.no_grad(): self.kinks[l] += (z.abs() < KINK_TOL).sum(0).numpy()
            x = ACT[a](z)
        if training: x = self._dropout(x, len(self.W), call)
        return x


def keras_loss(kind, y, p, sw):
    eps = 1e-7
    kind = kind.lower()
    if kind in ('categorical_crossentropy', 'cce'):
        p = p / p.sum(-1, keepdim=True)
        l = -(y * torch.log(p.clamp(eps, 1 - eps))).sum(-1)
    elif kind in ('binary_crossentropy', 'bce'):
        pc = p.clamp(eps, 1 - eps)
        l = -(y * torch.log(pc) + (1 - y) * torch.log(1 - pc)).mean(-1)
    elif kind in ('mse', 'mean_squared_error'):
        l = ((p - y) ** 2).mean(-1)
    elif kind in ('mae', 'mean_absolute_error'):
        l = (p - y).abs().mean(-1)
    else:
        raise ValueError(kind)
    return (l * sw).sum() / max(l.shape[0], 1)


def _sp(triple, dtype):
    idx, val, shp = triple
    idx = np.asarray(idx).reshape(-1, 2)
    return torch.sparse_coo_tensor(torch.from_numpy(np.ascontiguousarray(idx.T[[1, 0]])),
                                   torch.tensor(np.asarray(val, dtype=np.float64).reshape(-1), dtype=dtype),
                                   (int(shp[1]), int(shp[0]))).coalesce()


def _checkpointed(ns, comps):
    """state -> ns(concat(comps(state))) with the iteration's intermediates (the [N, in_dim] concatenation, its normalised copy, the
    pre-activations: ~6 GB per iteration at 1 M nodes in float64) recomputed in the backward pass instead of kept: the SAME float64
    operations in the same order (torch.utils.checkpoint), what lets the million-node oracle fit a host."""
    from torch.utils.checkpoint import checkpoint
    first = [True]
    def fn(state):
        if not first[0]: ns._recompute = True
        try:
            call0 = ns.calls
            out = ns(torch.cat(comps(state), dim=1))
            if not first[0]: ns.calls = call0              # (the recomputation is the same call, not a new one)
            return out
        finally:
            first[0] = False
            ns._recompute = False
    return lambda state: checkpoint(fn, state, use_reentrant=False)


def train_step(nodes, arcs, adjacency, arcnode, nodegraph, mask, *, net_state, net_output, state_vect_dim, max_iteration,
               state_threshold, focus, state0, y, sample_weight, loss, average_st_grads=False, dtype=torch.float64, seed=None,
               checkpoint_iterations=False):
    """Returns dict(k, loss, y_pred, grads_state, grads_output, moving_state, moving_output, kinks_state, kinks_output) as numpy.
    `seed`: the step's seed for the Dropout masks (the product derives its step key as mix32(0x5EED, seed)).
    `checkpoint_iterations`: keep only the states between iterations and recompute each iteration in the backward pass (large graphs)."""
    step_seed = mix32(0x5EED, int(seed)) if seed is not None else 0
    ns, no = Net(*net_state, dtype=dtype, net_id=0, step_seed=step_seed), Net(*net_output, dtype=dtype, net_id=1000, step_seed=step_seed)
    X = torch.tensor(np.asarray(nodes), dtype=dtype)
    lab = torch.tensor(np.asarray(arcs)[:, 2:], dtype=dtype)
    At, ANt = _sp(adjacency, dtype), _sp(arcnode, dtype)
    agg_arcs = torch.sparse.mm(ANt, lab) if lab.shape[1] else torch.zeros((X.shape[0], 0), dtype=dtype)
    d = state_vect_dim
    if d > 0:
        state = torch.tensor(np.asarray(state0), dtype=dtype)
        agg_nodes = torch.sparse.mm(At, X)
        comps = lambda s: [s, X, torch.sparse.mm(At, s), agg_nodes, agg_arcs]
    else:
        state = X.clone()
        comps = lambda s: [s, torch.sparse.mm(At, s), agg_arcs]
    state_old = torch.ones_like(state)
    k = 0
    if checkpoint_iterations and not state.requires_grad: state = state.clone().requires_grad_()      # (checkpoint wants a differentiable input)
    while True:
        with torch.no_grad():
            dist = torch.sqrt(torch.sum(torch.square(state - state_old), dim=1))
            norm = torch.sqrt(torch.sum(torch.square(state_old), dim=1))
            go_on = bool(torch.any(dist > state_threshold * norm)) and k < max_iteration
        if not go_on:
            break
        step = _checkpointed(ns, comps) if checkpoint_iterations else (lambda s_: ns(torch.cat(comps(s_), dim=1)))
        state, state_old, k = step(state), state.detach(), k + 1
    mask = torch.from_numpy(np.asarray(mask, dtype=bool))
    sc = torch.cat([state, X], dim=1) if d > 0 else state
    if focus == 'a':
        idx = torch.from_numpy(np.asarray(adjacency[0]).reshape(-1, 2).astype(np.int64))
        inp = torch.cat([sc[idx].reshape(lab.shape[0], 2 * sc.shape[1]), lab], dim=1)[mask]
    else:
        inp = sc[mask]
    out = no(inp)
    if focus == 'g':
        out = torch.sparse.mm(_sp(nodegraph, dtype), out)
    yt = torch.tensor(np.asarray(y), dtype=dtype)
    sw = torch.ones(yt.shape[0], dtype=dtype) if sample_weight is None else torch.tensor(np.asarray(sample_weight), dtype=dtype)
    L = keras_loss(loss, yt, out, sw) + ns.penalty() + no.penalty()        # regularization_losses=self.losses (GNN.py:286)
    params = ns.trainable() + no.trainable()
    grads = torch.autograd.grad(L, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
    n_s = len(ns.trainable())
    gs, go = grads[:n_s], grads[n_s:]
    if average_st_grads and k > 0: gs = [g / k for g in gs]
    npy = lambda t: t.detach().numpy()
    return dict(k=k, loss=float(L.detach()), y_pred=npy(out), state=npy(state), grads_state=[npy(g) for g in gs],
                grads_output=[npy(g) for g in go],
                moving_state=(npy(ns.moving_mean), npy(ns.moving_var)) if ns.bn else None,
                moving_output=(npy(no.moving_mean), npy(no.moving_var)) if no.bn else None,
                kinks_state=ns.kinks, kinks_output=no.kinks)


def adam_update(p, g, m, v, step, lr=0.001, b1=0.9, b2=0.999, eps=1e-7):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    alpha = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    return p - alpha * m / (np.sqrt(v) + eps), m, v


def composite_train_step(nodes, arcs, dim_node_label, type_mask, composite_adjacencies, adjacency, arcnode, nodegraph, mask,
                         *, net_state, net_output, state_vect_dim, max_iteration, state_threshold, focus, state0, y,
                         sample_weight, loss, average_st_grads=False, dtype=torch.float64, checkpoint_iterations=False):
    """CompositeGNNnodeBased.train_step (GNN/Models/CompositeGNN.py:275-304) with torch autograd: one state network per
    node type applied to the boolean-masked rows, scattered back and summed (CompositeGNN.py:223-232).
    `checkpoint_iterations`: keep only the states between iterations and recompute each iteration (its T concatenations, their normalised
    copies, the pre-activations) in the backward pass - the same float64 operations in the same order (large graphs: see _checkpointed)."""
    nets = [Net(*n, dtype=dtype) for n in net_state]
    no = Net(*net_output, dtype=dtype)
    X = torch.tensor(np.asarray(nodes), dtype=dtype)
    lab = torch.tensor(np.asarray(arcs)[:, 2:], dtype=dtype)
    dims = [int(v) for v in np.asarray(dim_node_label).reshape(-1)]
    tm = torch.from_numpy(np.asarray(type_mask, dtype=bool))                 # (T, N)
    At, ANt = _sp(adjacency, dtype), _sp(arcnode, dtype)
    agg_nodes = [torch.sparse.mm(_sp(ca, dtype), X[:, :dt]) for ca, dt in zip(composite_adjacencies, dims)]
    agg_arcs = torch.sparse.mm(ANt, lab) if lab.shape[1] else torch.zeros((X.shape[0], 0), dtype=dtype)
    agg_comp = torch.cat(agg_nodes + [agg_arcs], dim=1)
    state = torch.tensor(np.asarray(state0), dtype=dtype) if state_vect_dim > 0 else X.clone()
    state_old = torch.ones_like(state)
    k = 0

    def iteration(state):
        agg = torch.sparse.mm(At, state)
        new = torch.zeros_like(state)
        for dt, m_, net in zip(dims, tm, nets):
            if not bool(m_.any()): continue
            inp = torch.cat([X[:, :dt], state, agg, agg_comp], dim=1)[m_]
            full = torch.zeros_like(state)
            full[m_] = net(inp)
            new = new + full
        return new

    def checkpointed_iteration(state):
        from torch.utils.checkpoint import checkpoint
        first = [True]
        def fn(s_):
            calls0 = [n.calls for n in nets]
            if not first[0]:
                for n in nets: n._recompute = True
            try:
                out = iteration(s_)
                if not first[0]:
                    for n, c0 in zip(nets, calls0): n.calls = c0          # (the recomputation is the same call, not a new one)
                return out
            finally:
                first[0] = False
                for n in nets: n._recompute = False
        return checkpoint(fn, state, use_reentrant=False)

    if checkpoint_iterations and not state.requires_grad: state = state.clone().requires_grad_()      # (checkpoint wants a differentiable input)
    while True:
        with torch.no_grad():
            dist = torch.sqrt(torch.sum(torch.square(state - state_old), dim=1))
            norm = torch.sqrt(torch.sum(torch.square(state_old), dim=1))
            go_on = bool(torch.any(dist > state_threshold * norm)) and k < max_iteration
        if not go_on:
            break
        state, state_old, k = (checkpointed_iteration(state) if checkpoint_iterations else iteration(state)), state.detach(), k + 1
    mask = torch.from_numpy(np.asarray(mask, dtype=bool))
    if focus == 'a':
        idx = torch.from_numpy(np.asarray(adjacency[0]).reshape(-1, 2).astype(np.int64))
        inp = torch.cat([state[idx].reshape(lab.shape[0], 2 * state.shape[1]), lab], dim=1)[mask]
    else:
        inp = state[mask]
    out = no(inp)
    if focus == 'g':
        out = torch.sparse.mm(_sp(nodegraph, dtype), out)
    yt = torch.tensor(np.asarray(y), dtype=dtype)
    sw = torch.ones(yt.shape[0], dtype=dtype) if sample_weight is None else torch.tensor(np.asarray(sample_weight), dtype=dtype)
    L = keras_loss(loss, yt, out, sw)
    params = [p for n in nets for p in n.trainable()] + no.trainable()
    grads = torch.autograd.grad(L, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
    npy = lambda t: t.detach().numpy()
    out_grads, pos = [], 0
    for n in nets:
        cnt = len(n.trainable())
        g = grads[pos:pos + cnt]; pos += cnt
        if average_st_grads and k > 0: g = [x / k for x in g]
        out_grads.append([npy(x) for x in g])
    return dict(k=k, loss=float(L.detach()), y_pred=npy(out), state=npy(state), grads_state=out_grads,
                grads_output=[npy(g) for g in grads[pos:]],
                moving_state=[(npy(n.moving_mean), npy(n.moving_var)) if n.bn else None for n in nets],
                moving_output=(npy(no.moving_mean), npy(no.moving_var)) if no.bn else None,
                kinks_state=[n.kinks for n in nets], kinks_output=no.kinks)


def _homogeneous_forward(ns, no, X, lab, At, agg_arcs, d, max_iteration, state_threshold, state0, mask, focus, adjacency, NGt):
    """One GNN layer with autograd-tracked inputs X (labels) / lab (arc labels): (k, state, node/arc-level out, task out)."""
    if d > 0:
        state = state0
        agg_nodes = torch.sparse.mm(At, X)
        comps = lambda s: [s, X, torch.sparse.mm(At, s), agg_nodes, agg_arcs]
    else:
        state = X
        comps = lambda s: [s, torch.sparse.mm(At, s), agg_arcs]
    state_old = torch.ones_like(state)
    k = 0
    while True:
        dist = torch.sqrt(torch.sum(torch.square(state - state_old), dim=1))
        norm = torch.sqrt(torch.sum(torch.square(state_old), dim=1))
        if not (bool(torch.any(dist > state_threshold * norm)) and k < max_iteration):
            break
        state, state_old, k = ns(torch.cat(comps(state), dim=1)), state, k + 1
    sc = torch.cat([state, X], dim=1) if d > 0 else state
    if focus == 'a':
        idx = torch.from_numpy(np.asarray(adjacency[0]).reshape(-1, 2).astype(np.int64))
        inp = torch.cat([sc[idx].reshape(lab.shape[0], 2 * sc.shape[1]), lab], dim=1)[mask]
    else:
        inp = sc[mask]
    out = no(inp)
    task = torch.sparse.mm(NGt, out) if focus == 'g' else out
    return k, state, out, task


def lgnn_train_step(nodes, arcs, adjacency, arcnode, nodegraph, mask, *, layers, get_state, get_output, focus, state0s, y,
                    sample_weight, loss, training_mode, average_st_grads=False, dtype=torch.float64):
    """LGNN.train_step (GNN/Models/LGNN.py:252-287) with torch autograd across all layers.
    layers: list of dict(net_state, net_output, state_vect_dim, max_iteration, state_threshold)."""
    X0 = torch.tensor(np.asarray(nodes), dtype=dtype)
    arcs_t = torch.tensor(np.asarray(arcs), dtype=dtype)
    At, ANt = _sp(adjacency, dtype), _sp(arcnode, dtype)
    NGt = _sp(nodegraph, dtype) if focus == 'g' else None
    mask_t = torch.from_numpy(np.asarray(mask, dtype=bool))
    nets = [(Net(*l['net_state'], dtype=dtype), Net(*l['net_output'], dtype=dtype)) for l in layers]
    X, A_full = X0, arcs_t
    ks, outs = [], []
    for i, (l, (ns, no)) in enumerate(zip(layers, nets)):
        lab = A_full[:, 2:]
        agg_arcs = torch.sparse.mm(ANt, lab) if lab.shape[1] else torch.zeros((X.shape[0], 0), dtype=dtype)
        s0 = None if l['state_vect_dim'] == 0 else torch.tensor(np.asarray(state0s[i]), dtype=dtype)
        k, state, out, task = _homogeneous_forward(ns, no, X, lab, At, agg_arcs, l['state_vect_dim'], l['max_iteration'],
                                                   l['state_threshold'], s0, mask_t, focus, adjacency, NGt)
        ks.append(k); outs.append(task)
        if i < len(layers) - 1:                                     # update_graph (LGNN.py:175-214)
            nodeplus, arcplus = [], []
            if get_state: nodeplus.append(state)
            if get_output:
                scat = torch.zeros((len(mask_t), out.shape[1]), dtype=dtype)
                scat = scat.index_put((torch.nonzero(mask_t).reshape(-1),), out)
                (arcplus if focus == 'a' else nodeplus).append(scat)
            X = torch.cat(nodeplus + [X0], dim=1)
            A_full = torch.cat(arcplus + [arcs_t], dim=1)
    yt = torch.tensor(np.asarray(y), dtype=dtype)
    sw = torch.ones(yt.shape[0], dtype=dtype) if sample_weight is None else torch.tensor(np.asarray(sample_weight), dtype=dtype)
    if training_mode == 'parallel':
        L = torch.stack([keras_loss(loss, yt, o, sw) for o in outs]).mean()
    else:
        L = keras_loss(loss, yt, torch.stack(outs).mean(0), sw)
    params = [p for ns, no in nets for p in ns.trainable() + no.trainable()]
    grads = torch.autograd.grad(L, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
    npy = lambda t: t.detach().numpy()
    res, pos = [], 0
    for (ns, no), k in zip(nets, ks):
        a, b = len(ns.trainable()), len(no.trainable())
        gs_, go_ = grads[pos:pos + a], grads[pos + a:pos + a + b]; pos += a + b
        if average_st_grads and k > 0: gs_ = [g / k for g in gs_]
        res.append(([npy(g) for g in gs_], [npy(g) for g in go_]))
    return dict(k=ks, loss=float(L.detach()), outs=[npy(o) for o in outs], grads=res)


def _composite_forward(nets, no, X, lab, dims, tm, At, ANt, CAts, d, max_iteration, state_threshold, state0, mask, focus,
                       adjacency, NGt):
    """One composite GNN layer with autograd-tracked labels X: (k, state, node/arc-level out, task out)."""
    agg_nodes = [torch.sparse.mm(ca, X[:, :dt]) for ca, dt in zip(CAts, dims)]
    agg_arcs = torch.sparse.mm(ANt, lab) if lab.shape[1] else torch.zeros((X.shape[0], 0), dtype=X.dtype)
    agg_comp = torch.cat(agg_nodes + [agg_arcs], dim=1)
    state = state0 if d > 0 else X
    state_old = torch.ones_like(state)
    k = 0
    while True:
        dist = torch.sqrt(torch.sum(torch.square(state - state_old), dim=1))
        norm = torch.sqrt(torch.sum(torch.square(state_old), dim=1))
        if not (bool(torch.any(dist > state_threshold * norm)) and k < max_iteration):
            break
        agg = torch.sparse.mm(At, state)
        new = torch.zeros((X.shape[0], nets[0].W[-1].shape[1]), dtype=X.dtype)
        for dt, m_, net in zip(dims, tm, nets):
            if not bool(m_.any()): continue
            inp = torch.cat([X[:, :dt], state, agg, agg_comp], dim=1)[m_]
            new = new.index_put((torch.nonzero(m_).reshape(-1),), net(inp))
        state, state_old, k = new, state, k + 1
    if focus == 'a':
        idx = torch.from_numpy(np.asarray(adjacency[0]).reshape(-1, 2).astype(np.int64))
        inp = torch.cat([state[idx].reshape(lab.shape[0], 2 * state.shape[1]), lab], dim=1)[mask]
    else:
        inp = state[mask]
    out = no(inp)
    task = torch.sparse.mm(NGt, out) if focus == 'g' else out
    return k, state, out, task


def composite_lgnn_train_step(nodes, arcs, dim_node_label, type_mask, composite_adjacencies, adjacency, arcnode, nodegraph,
                              mask, *, layers, get_state, get_output, focus, state0s, y, sample_weight, loss,
                              training_mode, average_st_grads=False, dtype=torch.float64):
    """CompositeLGNN (GNN/Models/CompositeLGNN.py) joint train_step with torch autograd.
    layers: list of dict(net_state=[...per type], net_output, state_vect_dim, max_iteration, state_threshold)."""
    X0 = torch.tensor(np.asarray(nodes), dtype=dtype)
    arcs_t = torch.tensor(np.asarray(arcs), dtype=dtype)
    At, ANt = _sp(adjacency, dtype), _sp(arcnode, dtype)
    CAts = [_sp(ca, dtype) for ca in composite_adjacencies]
    NGt = _sp(nodegraph, dtype) if focus == 'g' else None
    tm = torch.from_numpy(np.asarray(type_mask, dtype=bool))
    mask_t = torch.from_numpy(np.asarray(mask, dtype=bool))
    dims = [int(v) for v in np.asarray(dim_node_label).reshape(-1)]
    nets = [([Net(*n, dtype=dtype) for n in l['net_state']], Net(*l['net_output'], dtype=dtype)) for l in layers]
    X, ks, outs = X0, [], []
    for i, (l, (nss, no)) in enumerate(zip(layers, nets)):
        s0 = None if l['state_vect_dim'] == 0 else torch.tensor(np.asarray(state0s[i]), dtype=dtype)
        k, state, out, task = _composite_forward(nss, no, X, arcs_t[:, 2:], dims, tm, At, ANt, CAts, l['state_vect_dim'],
                                                 l['max_iteration'], l['state_threshold'], s0, mask_t, focus, adjacency, NGt)
        ks.append(k); outs.append(task)
        if i < len(layers) - 1:
            nodeplus = []
            if get_state: nodeplus.append(state)
            if get_output:
                scat = torch.zeros((len(mask_t), out.shape[1]), dtype=dtype)
                nodeplus.append(scat.index_put((torch.nonzero(mask_t).reshape(-1),), out))
            plus = sum(t.shape[1] for t in nodeplus)
            X = torch.cat(nodeplus + [X0], dim=1)
            dims = [dt + plus for dt in dims]
    yt = torch.tensor(np.asarray(y), dtype=dtype)
    sw = torch.ones(yt.shape[0], dtype=dtype) if sample_weight is None else torch.tensor(np.asarray(sample_weight), dtype=dtype)
    if training_mode == 'parallel':
        L = torch.stack([keras_loss(loss, yt, o, sw) for o in outs]).mean()
    else:
        L = keras_loss(loss, yt, torch.stack(outs).mean(0), sw)
    params = [p for nss, no in nets for p in [q for n in nss for q in n.trainable()] + no.trainable()]
    grads = torch.autograd.grad(L, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
    npy = lambda t: t.detach().numpy()
    res, pos = [], 0
    for (nss, no), k in zip(nets, ks):
        a = sum(len(n.trainable()) for n in nss); b = len(no.trainable())
        gs_, go_ = grads[pos:pos + a], grads[pos + a:pos + a + b]; pos += a + b
        if average_st_grads and k > 0: gs_ = [g / k for g in gs_]
        res.append(([npy(g) for g in gs_], [npy(g) for g in go_]))
    return dict(k=ks, loss=float(L.detach()), outs=[npy(o) for o in outs], grads=res)
