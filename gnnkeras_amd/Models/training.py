"""train_step / fit for the homogeneous and composite GNNs on MI355X: back-propagation through the unrolled loop.

Mirror of the reference's `GNNnodeBased.train_step` (`GNN/Models/GNN.py:277-306`): forward with `training=True`
(BatchNormalization on batch statistics, moving averages updated on every call — i.e. k times per step for the state
network), `compiled_loss(y, y_pred, sample_weight)`, gradients w.r.t. `net_state` and `net_output` trainable variables
through every executed iteration (BPTT; the reference gets them from an eager `GradientTape`), optional division of the
state gradients by k (`average_st_grads`, `:295`), `optimizer.apply_gradients`.

There is no autograd here: the host walks the iterations backwards and calls hand-written device primitives
(`include/gnnloop.h`, "training building blocks"): transposed CSR aggregate, MFMA weight-gradient GEMMs with
deterministic two-stage reductions, activation / BatchNormalization / softmax / loss gradients, Adam / SGD updates.
Per-iteration tensors are *recomputed* in the backward sweep (aggregate, folded weights, hidden layers) from the stored
states, so the tape is just the k+1 state matrices and the BN batch statistics.

Algebra used for the first layer of a network, with y = a⊙x + c the training-mode BatchNormalization (a = γ·rstd,
c = β − μ·a; a = 1, c = 0 without BN), P = XᵀdZ and q = colsum(dZ):
    dW = a⊙P + c qᵀ,  db = q,  Σ_r dy = W q,  Σ_r dy⊙x = rowsum(W⊙P)  ⇒  dβ, dγ and the BN input-gradient moments
without ever materialising the N × in_dim concatenation or its gradient.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _native as nat
from ..sparse import SparseMatrix, CSRByDestination
from .MLP import Sequential, BN_EPSILON, BN_MOMENTUM


# ----------------------------------------------------------------------------------------------------------------------
# optimizers (tf.keras.optimizers defaults), state kept on the device
# ----------------------------------------------------------------------------------------------------------------------
class Adam:
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.iterations, self._slots = 0, {}

    def apply_gradients(self, grads_and_vars, gate=None):
        """Every variable in ONE launch (`gnn_adam_multi`); the pointer tables are rebuilt only when the set of tensors changes.
        `gate`: address of a device int32 - the launch changes nothing when it holds 0 (`gnn_train_step`'s validity word: the gradients of
        a step whose persistent backward launch failed never reach the weights; include/gnnloop.h, ABI 7)."""
        self.iterations += 1
        grads_and_vars = list(grads_and_vars)
        if not grads_and_vars: return
        key = tuple((g.data_ptr(), p.data_ptr()) for g, p in grads_and_vars)
        tab = self._slots.get('table')
        if tab is None or tab[0] != key:
            for g, p in grads_and_vars:
                if p.data_ptr() not in self._slots: self._slots[p.data_ptr()] = (torch.zeros_like(p), torch.zeros_like(p), p)
            n = len(grads_and_vars)
            arr = lambda vals: (C.c_void_p * n)(*vals)
            ms = [self._slots[p.data_ptr()][0] for _, p in grads_and_vars]; vs = [self._slots[p.data_ptr()][1] for _, p in grads_and_vars]
            tab = self._slots['table'] = (key, arr([p.data_ptr() for _, p in grads_and_vars]), arr([g.data_ptr() for g, _ in grads_and_vars]),
                                          arr([t.data_ptr() for t in ms]), arr([t.data_ptr() for t in vs]),
                                          (C.c_size_t * n)(*[p.numel() for _, p in grads_and_vars]), n, [g for g, _ in grads_and_vars])
        _, P, G, M, V, N, n, _keep = tab
        dev = grads_and_vars[0][1].device
        nat.check(nat.lib().gnn_adam_multi(P, G, M, V, N, n, float(self.learning_rate), float(self.beta_1), float(self.beta_2),
                                           float(self.epsilon), self.iterations, C.c_void_p(gate or 0), nat.current_stream(dev)))


class SGD:
    def __init__(self, learning_rate=0.01, momentum=0.0):
        self.learning_rate, self.momentum = learning_rate, momentum
        self.iterations, self._slots = 0, {}

    def apply_gradients(self, grads_and_vars, gate=None):
        self.iterations += 1
        lib = nat.lib()
        for g, p in grads_and_vars:
            vel = None
            if self.momentum:
                key = p.data_ptr()
                if key not in self._slots: self._slots[key] = (torch.zeros_like(p), p)
                vel = self._slots[key][0]
            nat.check(lib.gnn_sgd_step(nat.ptr(p), nat.ptr(g), nat.ptr(vel), p.numel(), float(self.learning_rate),
                                       float(self.momentum), C.c_void_p(gate or 0), nat.current_stream(p.device)))


def get_optimizer(opt):
    if opt is None: raise RuntimeError('compile() the model with an optimizer before fit()')
    if isinstance(opt, str):
        name = opt.lower()
        if name == 'adam': return Adam()
        if name == 'sgd': return SGD()
        raise ValueError(f'unknown optimizer {opt!r}')
    if not hasattr(opt, 'apply_gradients'): raise TypeError('optimizer must provide apply_gradients(grads_and_vars)')
    return opt


# ----------------------------------------------------------------------------------------------------------------------
# thin wrappers over the device primitives; a "segment" is (2-D float32 view with stride(1) == 1, row-index or None)
# ----------------------------------------------------------------------------------------------------------------------
class _Prim:
    def __init__(self, device):
        self.dev, self.lib = device, nat.lib()
        self._ws = None
        self._stream = nat.current_stream(device)   # one lookup per step (a _Prim lives for one forward + backward)
        self._nbytes = {}

    def stream(self):
        return self._stream

    def _ws_bytes(self, fn, *dims):
        key = (fn, dims)
        nb = self._nbytes.get(key)
        if nb is None:
            nb = self._nbytes[key] = getattr(self.lib, fn)(*dims)
        return nb

    def ws(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.dev)
        return self._ws

    def new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def zeros(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.dev)

    def dense(self, segs, W, H, bias, act, Y, wrows=None, out_rowidx=None, center=None):
        """Y[:, :H] = act(sum_s (seg_s - center[wrow_s : wrow_s + width_s]) . W[wrow_s : wrow_s + width_s, :H] + bias).  W: 2-D view
        (ld = stride(0)); `center` (by weight row, or None): subtracted as the inputs are staged - the consumer of `fold(centred=True)`."""
        d = nat.DenseArgs()
        M = Y.shape[0] if out_rowidx is None else len(out_rowidx)
        d.M, d.H, d.n_segments = M, H, len(segs)
        off = 0
        for i, (x, ridx) in enumerate(segs):
            d.seg_ptr[i] = x.data_ptr(); d.seg_rowidx[i] = 0 if ridx is None else ridx.data_ptr()
            d.seg_ld[i], d.seg_width[i] = x.stride(0), x.shape[1]
            d.seg_wrow[i] = off if wrows is None else wrows[i]
            off += x.shape[1]
        d.W, d.ldw = W.data_ptr(), W.stride(0)
        d.bias = 0 if bias is None else bias.data_ptr()
        d.activation = act
        d.Y, d.ldy = Y.data_ptr(), Y.stride(0)
        d.out_rowidx = 0 if out_rowidx is None else out_rowidx.data_ptr()
        d.in_center = 0 if center is None else center.data_ptr()
        d.stream = self.stream()
        nat.check(self.lib.gnn_dense(C.byref(d)))
        return Y

    def aggregate(self, csr, X, F, out):
        c = nat.make_csr(csr)
        nat.check(self.lib.gnn_aggregate(C.byref(c), nat.ptr(X), X.stride(0), F, nat.ptr(out), out.stride(0), self.stream()))
        return out

    def fold(self, W, b, bn, mean, var, Wf, bf, centred=False):
        """Wf = a W, bf = b + (beta - mean a) W; `centred`: bf = b + beta W (the consumer subtracts the mean from its inputs: dense(center=))."""
        K, H = W.shape
        g, be = (bn[0], bn[1]) if bn is not None else (None, None)
        nat.check(self.lib.gnn_fold_bn(nat.ptr(W), nat.ptr(b), K, H, nat.ptr(g), nat.ptr(be), nat.ptr(mean), nat.ptr(var),
                                       BN_EPSILON, nat.ptr(Wf), nat.ptr(bf), int(bool(centred)), self.stream()))

    def colstats(self, x, ridx, M, mean, var):
        K = x.shape[1]
        nb = self._ws_bytes('gnn_colstats_workspace_bytes', K, M)
        ws = self.ws(nb)
        nat.check(self.lib.gnn_colstats(nat.ptr(x), x.stride(0), nat.ptr(ridx), K, M, nat.ptr(mean), nat.ptr(var), None, None,
                                        BN_MOMENTUM, None, nat.ptr(ws), ws.numel(), self.stream()))

    def dense_grad(self, x, ridx, dZ, M, P, q, accumulate, center=None):
        """P (+)= (x[rows] - center)^T dZ, q (+)= colsum(dZ); `center`: [K] view or None."""
        K, H = x.shape[1], dZ.shape[1]
        assert P.is_contiguous() and P.shape[-1] == H
        nb = self._ws_bytes('gnn_dense_grad_workspace_bytes', K, H, M)
        ws = self.ws(nb)
        nat.check(self.lib.gnn_dense_grad(nat.ptr(x), x.stride(0), nat.ptr(ridx), K, nat.ptr(dZ), dZ.stride(0), H, M,
                                          nat.ptr(P), nat.ptr(q), int(accumulate), nat.ptr(center), nat.ptr(ws), ws.numel(), self.stream()))

    def act_grad(self, G, Y, dZ, act):
        M, H = Y.shape
        nat.check(self.lib.gnn_act_grad(nat.ptr(G), G.stride(0), nat.ptr(Y), Y.stride(0), nat.ptr(dZ), dZ.stride(0), M, H, act,
                                        self.stream()))
        return dZ

    def first_layer_param_grads(self, P, q, W, bn, mean, var, M, dW, db, dgamma, dbeta, m1, m2, accumulate, centered=False):
        """`centered`: P was formed from inputs with the BatchNormalization column means already subtracted (dense_grad(center=))."""
        K, H = W.shape
        g, be = (bn[0], bn[1]) if bn is not None else (None, None)
        nat.check(self.lib.gnn_first_layer_param_grads(nat.ptr(P), nat.ptr(q), nat.ptr(W), K, H, nat.ptr(g), nat.ptr(be),
                                                       nat.ptr(mean), nat.ptr(var), BN_EPSILON, M, nat.ptr(dW), nat.ptr(db),
                                                       nat.ptr(dgamma), nat.ptr(dbeta), nat.ptr(m1), nat.ptr(m2),
                                                       int(accumulate), int(bool(centered and bn is not None)), self.stream()))

    def bn_input_grad(self, dy, x, ridx, M, k0, bn, mean, var, m1, m2, dx):
        width = dy.shape[1]
        g = bn[0] if bn is not None else None
        nat.check(self.lib.gnn_bn_input_grad(nat.ptr(dy), dy.stride(0), nat.ptr(x), x.stride(0), nat.ptr(ridx), M, width, k0,
                                             nat.ptr(g), nat.ptr(mean), nat.ptr(var), BN_EPSILON, nat.ptr(m1), nat.ptr(m2),
                                             nat.ptr(dx), dx.stride(0), self.stream()))

    def dropout(self, x, y, rate, key, alpha, backward):
        """Keras Dropout / AlphaDropout with the mask of `key` (forward value, or backward gradient); x and y may alias."""
        M, H = x.shape
        nat.check(nat.lib().gnn_dropout(nat.ptr(x), x.stride(0), nat.ptr(y), y.stride(0), M, H, float(rate), int(key) & 0xFFFFFFFF,
                                        int(bool(alpha)), int(bool(backward)), self.stream()))
        return y

    def scatter_add_rows(self, D, idx, G):
        nat.check(self.lib.gnn_scatter_add_rows(nat.ptr(D), D.stride(0), nat.ptr(idx), D.shape[0], D.shape[1], nat.ptr(G),
                                                G.stride(0), self.stream()))

    def axpby(self, a, x, b, y, out):
        nat.check(self.lib.gnn_axpby(float(a), nat.ptr(x), float(b), nat.ptr(y), nat.ptr(out), x.numel(), self.stream()))

    def converged_gated(self, state, state_old, thr, gate, flag, k_dev, k_val):
        """flag |= any_row(||state - state_old|| > thr ||state_old||) when *gate != 0 (gate None: always); k_dev = k_val then."""
        N, S = state.shape
        nat.check(self.lib.gnn_converged_gated(nat.ptr(state), nat.ptr(state_old), N, S, state.stride(0), float(thr), nat.ptr(gate),
                                               nat.ptr(flag), nat.ptr(k_dev), float(k_val), self.stream()))

    def loss_grad(self, kind, y, y_pred, sw, dpred, loss_rows):
        R, T = y_pred.shape
        nat.check(self.lib.gnn_loss_grad(kind, nat.ptr(y), nat.ptr(y_pred), nat.ptr(sw), R, T, nat.ptr(dpred), nat.ptr(loss_rows),
                                         self.stream()))


class _NetGrads:
    """Gradient buffers of one Sequential in `trainable_variables` order (BN: γ, β; Dense: W, b ...)."""

    def __init__(self, net: Sequential, prim: _Prim, grads: bool = True):
        self.net = net
        self.bn = net.batch_normalization
        w = net.weights
        self.bn_params = (w[0], w[1]) if self.bn else None          # γ, β
        self.moving = (w[2], w[3]) if self.bn else None
        base = 4 if self.bn else 0
        self.W = [w[base + 2 * i] for i in range(len(net.units))]
        self.b = [w[base + 2 * i + 1] for i in range(len(net.units))]
        self.acts = [nat.ACTIVATIONS[a] for a in net.activations]
        self.dgamma = prim.zeros(net.input_dim) if self.bn and grads else None
        self.dbeta = prim.zeros(net.input_dim) if self.bn and grads else None
        self.dW = [torch.zeros_like(x) for x in self.W] if grads else None
        self.db = [torch.zeros_like(x) for x in self.b] if grads else None
        self.touched = False                                        # False until the first accumulation of this step
        # Dropout layers by position in the Dense list (reference MLP.py:60-66: position p = in front of Dense p, p = number of
        # Dense layers = behind the last one): {p: [(rate, index of the dropout layer), ...]}
        self.drop, self.alpha, self.net_id = {}, bool(getattr(net, 'alphadropout', False)), 0
        pos = [int(v) for v in (net.dropout_pos or [])]
        if pos != sorted(pos): raise ValueError('dropout_pos must be ascending (the reference inserts the layers in that order)')
        for i, (r, q) in enumerate(zip(net.dropout_rate or [], pos)):
            if not 0 <= q <= len(net.units): raise ValueError(f'dropout_pos {q} outside [0, {len(net.units)}]')
            if float(r) > 0: self.drop.setdefault(q, []).append((float(r), i))

    @classmethod
    def forward_only(cls, net: Sequential):
        """The same view of a network without gradient buffers (standalone training-mode calls)."""
        return cls(net, None, grads=False)

    def variables(self):
        v = list(self.bn_params) if self.bn else []
        for W, b in zip(self.W, self.b): v += [W, b]
        return v

    def gradients(self):
        g = [self.dgamma, self.dbeta] if self.bn else []
        for W, b in zip(self.dW, self.db): g += [W, b]
        return g


class _Acts(list):
    """Activations of one network call: the list of Dense outputs + `.inputs` (what each next layer consumed), `.out`, `.call`."""
    inputs, out, call = None, None, 0


def _lowbias32(h):
    h &= 0xFFFFFFFF
    h ^= h >> 16; h = (h * 0x7feb352d) & 0xFFFFFFFF
    h ^= h >> 15; h = (h * 0x846ca68b) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def _eye(ng, K, p):
    """Identity [K, K] on the primitives' device, cached on the network's gradient holder (dropout_pos = 0)."""
    e = getattr(ng, '_eye', None)
    if e is None or e.shape[0] != K:
        e = ng._eye = p.zeros(K, K)
        e.fill_diagonal_(1.0)
    return e


def _mix32(*values):
    """Key of a Dropout call from small integers (step seed, network, call, layer): the device hashes (key, row, column)."""
    h = 0x9E3779B9
    for v in values: h = _lowbias32(h ^ (int(v) & 0xFFFFFFFF))
    return h


class LoopTrainer:
    """One instance per model; owns gradient buffers and scratch. Homogeneous node / arc / graph focus."""

    prim_cls = _Prim                # the device primitives (tests/test_data_parallel.py swaps in a NumPy stand-in on CPU)

    def __init__(self, model, dp=None):
        """`dp`: a `gnnkeras_amd.data_parallel.DPContext` - the batch is then ONE SHARD (a subset of the graphs) of a merged batch
        whose other shards run on the other ranks, and the step reproduces the single-process step on the merged batch: BatchNorm
        statistics, the convergence test, the loss normalisation and the weight-gradient sums span all ranks."""
        self.model = model
        self.dp = dp

    # ---- generic MLP forward (training mode) / backward over segmented inputs -------------------------------------------
    def _drop_key(self, ng: _NetGrads, call, index):
        """32-bit key of one Dropout layer call: step seed, network, call number (iteration), layer index."""
        return _mix32(self.drop_seed, ng.net_id, int(call), int(index))

    def _dropped(self, ng: _NetGrads, q, x, call):
        """What the layer behind position q consumes: x through the Dropout layers sitting there (a copy), or x itself."""
        if q not in ng.drop: return x
        y = x
        for rate, index in ng.drop[q]:
            y = self.prim.dropout(y, self.prim.new(*x.shape) if y is x else y, rate, self._drop_key(ng, call, index), ng.alpha, False)
        return y

    def _mlp_forward(self, ng: _NetGrads, segs, M, stats=None, const_stats=None, call=0):
        """segs: [(view, rowidx)] in input-column order. Returns (hs, (mean, var)|None). `stats` given => reuse (backward
        recompute); else batch statistics are computed (constant segments may come from `const_stats`: {index: (mean, var)}).
        `hs` holds every Dense layer's activation; `hs.inputs[l]` is what layer l + 1 consumed (the same array, or its
        dropped-out copy) and `hs.out` the network's output.  `call` numbers the calls of this network within the step: it
        keys the Dropout masks, so the backward recompute of call t sees the masks of the forward call t."""
        p, net = self.prim, ng.net
        mean = var = None
        if M == 0 and self.dp is not None:
            # data-parallel step, a network without a single row on THIS rank (a node type that lives on other shards only): the rank
            # still takes part in the collective that forms the merged batch's statistics - with weight 0 - and keeps them for the
            # moving averages and the backward sweep (every rank applies the same updates)
            if ng.bn and stats is None:
                z = p.zeros(net.input_dim)
                mean, var = self.dp.combine_stats(z, z.clone(), getattr(ng, 'dp_kind', 'nodes'))
            elif ng.bn:
                mean, var = stats
            return None, ((mean, var) if ng.bn else None)
        if ng.bn:
            if stats is not None:
                mean, var = stats
            else:
                mean, var = p.new(net.input_dim), p.new(net.input_dim)
                off = 0
                for i, (x, ridx) in enumerate(segs):
                    w = x.shape[1]
                    if const_stats is not None and i in const_stats:
                        mean[off:off + w].copy_(const_stats[i][0]); var[off:off + w].copy_(const_stats[i][1])
                    else:
                        p.colstats(x, ridx, M, mean[off:off + w], var[off:off + w])
                    off += w
                if self.dp is not None: mean, var = self.dp.combine_stats(mean, var, getattr(ng, 'dp_kind', 'out' if ng is self.go else 'nodes'))   # statistics of the MERGED batch
        drop0 = 0 in ng.drop
        if drop0:
            # Dropout in front of the first Dense (reference MLP.py:60-66 with position 0; BatchNormalization is inserted in front of it,
            # :68-71): x -> BN -> Dropout -> Dense 0.  The normalised input is materialised as the output of an IDENTITY Dense with
            # the statistics folded in ([M, input_dim], the virtual concatenation made real), dropped out, and Dense 0 runs plain.
            K = net.input_dim
            eye = _eye(ng, K, p)
            if ng.bn:
                Wn, bn_ = p.new(K, K), p.new(K)
                p.fold(eye, p.zeros(K), ng.bn_params, mean, var, Wn, bn_, centred=True)
            else:
                Wn, bn_ = eye, None
            xn = p.dense(segs, Wn, K, bn_, 0, p.new(M, K), center=mean if ng.bn else None)
            x0 = self._dropped(ng, 0, xn, call)
            hs = _Acts([p.dense([(x0, None)], ng.W[0], net.units[0], ng.b[0], ng.acts[0], p.new(M, net.units[0]))])
            hs.x0 = x0
        else:
            # BatchNormalization as a (x - mean) + beta: the column mean leaves the value as it is staged (a folded bias b + (beta - mean a) W
            # would cancel mean a W against a x W afterwards - digits lost in proportion to mean / sigma: labels far from zero)
            if ng.bn:
                Wf, bf = p.new(*ng.W[0].shape), p.new(ng.W[0].shape[1])
                p.fold(ng.W[0], ng.b[0], ng.bn_params, mean, var, Wf, bf, centred=True)
            else:
                Wf, bf = ng.W[0], ng.b[0]
            hs = _Acts([p.dense(segs, Wf, net.units[0], bf, ng.acts[0], p.new(M, net.units[0]), center=mean if ng.bn else None)])
        hs.inputs = [self._dropped(ng, 1, hs[0], call)]
        for l in range(1, len(net.units)):
            hs.append(p.dense([(hs.inputs[-1], None)], ng.W[l], net.units[l], ng.b[l], ng.acts[l], p.new(M, net.units[l])))
            hs.inputs.append(self._dropped(ng, l + 1, hs[l], call))
        hs.out, hs.call = hs.inputs[-1], call
        return hs, ((mean, var) if ng.bn else None)

    def _mlp_backward(self, ng: _NetGrads, segs, hs, G, M, stats, dx_requests):
        """G = dL/d hs.out (overwritten). dx_requests: [(segment index, out view [M, width])]."""
        p, net = self.prim, ng.net
        acc = ng.touched
        if M == 0 and self.dp is not None:
            # (see _mlp_forward: no rows of this network here - zeros into the all-reduced sums, the same parameter gradients everywhere)
            K, H = ng.W[0].shape
            P, q = p.zeros(K, H), p.zeros(H)
            mean, var = stats if stats is not None else (None, None)
            m1 = m2 = None
            if ng.bn: m1, m2 = p.new(K), p.new(K)
            self.dp.all_reduce_sum(P, q)
            p.first_layer_param_grads(P, q, ng.W[0], ng.bn_params, mean, var, self.dp.total_rows(getattr(ng, 'dp_kind', 'nodes')), ng.dW[0], ng.db[0],
                                      ng.dgamma, ng.dbeta, m1, m2, acc, centered=bool(ng.bn and mean is not None))
            ng.touched = True
            return

        def through_dropout(q, G_):                                # d loss / d (input of the Dropout layers at position q)
            for rate, index in reversed(ng.drop.get(q, [])):
                p.dropout(G_, G_, rate, self._drop_key(ng, hs.call, index), ng.alpha, True)
            return G_
        G = through_dropout(len(net.units), G)
        for l in range(len(net.units) - 1, 0, -1):
            dZ = p.act_grad(G, hs[l], G, ng.acts[l])
            p.dense_grad(hs.inputs[l - 1], None, dZ, M, ng.dW[l], ng.db[l], acc)
            Wt = ng.W[l].t().contiguous()
            G = through_dropout(l, p.dense([(dZ, None)], Wt, net.units[l - 1], None, 0, p.new(M, net.units[l - 1])))
        dZ = p.act_grad(G, hs[0], G, ng.acts[0])
        K, H = ng.W[0].shape
        drop0 = 0 in ng.drop
        if drop0:
            # Dense 0 is a plain layer over the dropped-out input; the gradient then passes the Dropout masks and meets the
            # identity Dense that carries BatchNormalization: the same first-layer algebra with W = I (S1 = q, S2 = diag(P))
            p.dense_grad(hs.x0, None, dZ, M, ng.dW[0], ng.db[0], acc)
            if self.dp is not None: raise NotImplementedError('dropout_pos = 0 under data parallelism')
            dZ = through_dropout(0, p.dense([(dZ, None)], ng.W[0].t().contiguous(), K, None, 0, p.new(M, K)))
            H = K
            if not ng.bn:
                ng.touched = True
                offs = np.cumsum([0] + [x.shape[1] for x, _ in segs])
                for si, out in dx_requests or []:
                    out.copy_(dZ[:, int(offs[si]):int(offs[si]) + segs[si][0].shape[1]])
                return
        P, q = p.new(K, H), p.new(H)
        mean, var = stats if stats is not None else (None, None)
        centred = bool(ng.bn and mean is not None)     # P = (X - mean)^T dZ: formed from centred rows, nothing of the mean's size to cancel later
        off = 0
        for i, (x, ridx) in enumerate(segs):
            w = x.shape[1]
            p.dense_grad(x, ridx, dZ, M, P[off:off + w], q if i == 0 else None, False, center=mean[off:off + w] if centred else None)
            off += w
        m1 = m2 = None
        if ng.bn: m1, m2 = p.new(K), p.new(K)
        M_all = M
        if drop0:
            eye = _eye(ng, K, p)
            p.first_layer_param_grads(P, q, eye, ng.bn_params, mean, var, M_all, p.zeros(K, K), p.zeros(K), ng.dgamma, ng.dbeta, m1, m2, acc,
                                      centered=centred)
            ng.touched = True
            offs = np.cumsum([0] + [x.shape[1] for x, _ in segs])
            for si, out in dx_requests or []:
                x, ridx = segs[si]
                k0, w = int(offs[si]), x.shape[1]
                out.copy_(dZ[:, k0:k0 + w])
                p.bn_input_grad(out, x, ridx, M, k0, ng.bn_params, mean, var, m1, m2, out)
            return
        if self.dp is not None:
            # P = X^T dZ and q = colsum(dZ) over the rows of EVERY shard: the first layer's parameter gradients and the BatchNorm
            # input-gradient moments m1 / m2 (means over all rows of the merged batch) follow from the sums
            self.dp.all_reduce_sum(P, q)
            M_all = self.dp.total_rows(getattr(ng, 'dp_kind', 'out' if ng is self.go else 'nodes'))
        p.first_layer_param_grads(P, q, ng.W[0], ng.bn_params, mean, var, M_all, ng.dW[0], ng.db[0], ng.dgamma, ng.dbeta,
                                  m1, m2, acc, centered=centred)
        ng.touched = True
        if dx_requests:
            Wt = ng.W[0].t().contiguous()                          # [H, K]
            offs = np.cumsum([0] + [x.shape[1] for x, _ in segs])
            for si, out in dx_requests:
                x, ridx = segs[si]
                k0, w = int(offs[si]), x.shape[1]
                p.dense([(dZ, None)], Wt[:, k0:k0 + w], w, None, 0, out)
                p.bn_input_grad(out, x, ridx, M, k0, ng.bn_params, mean, var, m1, m2, out)

    # ---- training-mode forward: records the tape ------------------------------------------------------------------------
    def forward(self, x_list, state0=None, seed=None, node_level=False):
        """Training-mode forward of one batch. Returns the tape (a `types.SimpleNamespace`) holding everything the
        backward sweep needs: the k+1 states, the BN batch statistics per iteration, the output-network activations.
        `node_level=True` makes a graph-focused model stop at the per-node outputs (what LGNN feeds to the next layer,
        reference LGNN.py:225-246 uses `GNNnodeBased.Loop` for that)."""
        from types import SimpleNamespace
        m = self.model
        tp = SimpleNamespace()
        composite = tp.composite = isinstance(m.net_state, (list, tuple))
        inputs = m.process_inputs(x_list)
        if composite:
            nodes, arcs, dim_node_label, type_mask, set_mask, output_mask, _cas, adjacency, arcnode, nodegraph = inputs
        else:
            nodes, arcs, dim_node_label, set_mask, output_mask, adjacency, arcnode, nodegraph = inputs
        nat.require_device(nodes, 'nodes')
        dev = tp.dev = nodes.device
        self.prim = p = tp.p = self.prim_cls(dev)
        nodes = tp.nodes = nodes.to(torch.float32).contiguous(); arcs = arcs.to(torch.float32).contiguous()
        N, L = nodes.shape
        A = arcs.shape[1] - 2
        d = m.state_vect_dim
        S = d if d > 0 else L
        tp.N, tp.L, tp.A, tp.d, tp.S = N, L, A, d, S
        focus = tp.focus = 'n' if (node_level and m._focus == 'g') else m._focus
        tp.pooled = m._focus == 'g' and not node_level
        nets_s = list(m.net_state) if composite else [m.net_state]
        for n_ in nets_s: n_.to(dev)
        m.net_output.to(dev)
        gs, go = [_NetGrads(n_, p) for n_ in nets_s], _NetGrads(m.net_output, p)
        for i, g_ in enumerate(gs): g_.net_id, g_.dp_kind = i, (f'type{i}' if composite else 'nodes')      # (dp_kind: whose rows the network sees, for the data-parallel counts)
        go.net_id, go.dp_kind = 1000, 'out'
        # Dropout masks of this step: `seed` when given (reproducible steps, tests), else a per-model step counter
        m._dropout_step = getattr(m, '_dropout_step', 0) + 1
        self.drop_seed = tp.drop_seed = _mix32(0x5EED, int(seed)) if seed is not None else _mix32(id(m) & 0xFFFFFFFF, m._dropout_step)
        tp.gs, tp.go = gs, go
        self.gs, self.go = (gs if composite else gs[0]), go
        adj = tp.adj = adjacency.device_csr(dev)
        tp.adj_src = _by_source(adjacency, dev)
        an = arcnode.device_csr(dev)
        tp.arcnode = arcnode
        tp.nodegraph = nodegraph
        tp.cas = _cas if composite else None
        from .GNN import _squeeze_last, _arc_endpoints
        out_index = tp.out_index = m._out_index(_squeeze_last(set_mask).to(dev), _squeeze_last(output_mask).to(dev))
        M = tp.M = len(out_index)

        # node types: row lists (None = all rows for the homogeneous model)
        if composite:
            dims = [int(v) for v in (dim_node_label.reshape(-1).tolist() if isinstance(dim_node_label, torch.Tensor)
                                     else np.asarray(dim_node_label).reshape(-1))]
            type_nodes, offsets = m._type_lists(_squeeze_last(type_mask).to(dev))
            rows = [type_nodes[int(offsets[t]):int(offsets[t + 1])].contiguous() for t in range(len(dims))]
        else:
            dims, rows = [L], [None]
        tp.dims, tp.rows = dims, rows
        T_types = tp.T_types = len(rows)
        counts = tp.counts = [N if r is None else len(r) for r in rows]
        rows_long = tp.rows_long = [None if r is None else r.long() for r in rows]

        # ---- setup aggregates (GNN.py:254-258 / CompositeGNN.py:251-253) ----
        arc_labels = tp.arc_labels = arcs[:, 2:]
        agg_arcs = p.aggregate(an, arc_labels, A, p.new(N, max(A, 1)))[:, :A] if A > 0 else None
        agg_comp = agg_nodes = None
        if composite:
            sum_d = sum(dims)
            agg_comp = p.new(N, max(sum_d + A, 1))                  # [agg_nodes_0 | ... | agg_nodes_{T-1} | agg_arcs]
            col = 0
            for t, ca in enumerate(_cas):
                if dims[t] > 0: p.aggregate(ca.device_csr(dev), nodes[:, :dims[t]], dims[t], agg_comp[:, col:col + dims[t]])
                col += dims[t]
            if A > 0: agg_comp[:, col:col + A].copy_(agg_arcs)
            agg_comp = agg_comp[:, :sum_d + A]
        else:
            agg_nodes = p.aggregate(adj, nodes, L, p.new(N, L)) if d > 0 else None
        if d > 0:
            if state0 is None:
                gen = None
                if seed is not None:
                    gen = torch.Generator(device=dev); gen.manual_seed(int(seed))
                state0 = torch.empty((N, d), device=dev, dtype=torch.float32).normal_(0.0, 0.1, generator=gen)      # tf.random.normal(stddev=0.1), one launch
            s_init = state0.to(dev, torch.float32)
        else:
            s_init = nodes
        K_it = m.max_iteration
        states = tp.states = p.new(K_it + 1, N, S)
        states[0].copy_(s_init)
        agg = tp.agg = p.new(N, S)

        def state_segs(t, ty):
            """(segments, index of the state segment, of the agg segment, of the label segment, of the agg-label segment)"""
            r = rows[ty]
            if composite:                                           # [labels[:, :d_t] | state | agg_state | agg_component]
                segs = [(nodes[:, :dims[ty]], r)] if dims[ty] > 0 else []
                i_state = len(segs)
                segs += [(states[t], r), (agg, r)]
                if agg_comp.shape[1] > 0: segs.append((agg_comp, r))
                return segs, i_state, i_state + 1, (0 if dims[ty] > 0 else None), (i_state + 2 if agg_comp.shape[1] > 0 else None)
            segs = [(states[t], None)]                              # [state | labels | agg_state | agg_labels | agg_arcs]
            if d > 0: segs.append((nodes, None))
            segs.append((agg, None))
            if d > 0: segs.append((agg_nodes, None))
            if A > 0: segs.append((agg_arcs, None))
            return segs, 0, (2 if d > 0 else 1), (1 if d > 0 else None), (3 if d > 0 else None)
        tp.state_segs = state_segs

        const_stats = []
        for ty in range(T_types):
            cs = None
            if gs[ty].bn and counts[ty] > 0:
                cs = {}
                segs, i_state, i_agg, _, _ = state_segs(0, ty)
                for i, (xv, ridx) in enumerate(segs):
                    if i in (i_state, i_agg): continue
                    mu, va = p.new(xv.shape[1]), p.new(xv.shape[1])
                    p.colstats(xv, ridx, counts[ty], mu, va)
                    cs[i] = (mu, va)
            const_stats.append(cs)

        # ---- every iteration is computed, the predicate only records where the loop stops ----
        flags = torch.zeros(K_it + 2, dtype=torch.int32, device=dev)
        k_dev = torch.zeros((), dtype=torch.float32, device=dev)
        dp = self.dp
        if dp is not None: dp.begin_step(N, M, type_counts=counts if composite else None)
        # rows of every network over ALL shards (data parallel: a type may have rows on other ranks only)
        all_counts = tp.all_counts = [dp.total_rows(gs[ty].dp_kind) for ty in range(T_types)] if dp is not None else list(counts)
        p.converged_gated(states[0], None, m.state_threshold, None, flags[0:1], None, 0.0)
        if dp is not None: dp.any_flag(flags[0:1])                    # `reduce_any` over the nodes of the merged batch (GNN.py:212)
        stats_t = tp.stats_t = []
        for t in range(K_it):
            p.aggregate(adj, states[t], S, agg)
            st_t = []
            for ty in range(T_types):
                if all_counts[ty] == 0:
                    st_t.append(None); continue
                segs = state_segs(t, ty)[0]
                hs, st = self._mlp_forward(gs[ty], segs, counts[ty], const_stats=const_stats[ty], call=t)
                if hs is None: pass                                  # (no rows of this type on this rank)
                elif rows[ty] is None: states[t + 1].copy_(hs.out)
                else: states[t + 1].index_copy_(0, rows_long[ty], hs.out)
                st_t.append(st)
            stats_t.append(st_t)
            p.converged_gated(states[t + 1], states[t], m.state_threshold, flags[t:t + 1], flags[t + 1:t + 2], k_dev, t + 1)
            if dp is not None: dp.any_flag(flags[t + 1:t + 2])
        k = tp.k = int(float(k_dev))                                # the one host synchronisation of the step
        for ty in range(T_types):                                   # k moving-average updates, applied in order (closed form)
            if gs[ty].bn and k > 0 and all_counts[ty] > 0:
                mm, mv = gs[ty].moving
                wts = torch.tensor([BN_MOMENTUM ** (k - 1 - t) * (1 - BN_MOMENTUM) for t in range(k)], device=dev)
                mm.mul_(BN_MOMENTUM ** k).add_((torch.stack([stats_t[t][ty][0] for t in range(k)]) * wts[:, None]).sum(0))
                mv.mul_(BN_MOMENTUM ** k).add_((torch.stack([stats_t[t][ty][1] for t in range(k)]) * wts[:, None]).sum(0))
        state_k = tp.state = states[k]

        # ---- output network (training mode) ----
        with_labels = tp.with_labels = (not composite) and d > 0    # composite filters use the state only (CompositeGNN.py:237-239)
        if focus == 'a':
            es, ed = _arc_endpoints(adjacency, dev)
            tp.isrc, tp.idst = es[out_index.long()].contiguous(), ed[out_index.long()].contiguous()
            osegs = []
            for ends in (tp.isrc, tp.idst):
                osegs.append((state_k, ends))
                if with_labels: osegs.append((nodes, ends))
            if A > 0: osegs.append((arc_labels, out_index))
        else:
            osegs = [(state_k, out_index)]
            if with_labels: osegs.append((nodes, out_index))
        tp.osegs = osegs
        T = tp.T = m.net_output.units[-1]
        if M > 0:
            tp.ohs, tp.ostats = self._mlp_forward(go, osegs, M)
            if go.bn:
                mm, mv = go.moving
                mm.mul_(BN_MOMENTUM).add_(tp.ostats[0] * (1 - BN_MOMENTUM)); mv.mul_(BN_MOMENTUM).add_(tp.ostats[1] * (1 - BN_MOMENTUM))
            tp.out_nodes = tp.ohs.out
        else:
            tp.ohs, tp.ostats, tp.out_nodes = None, None, p.new(0, T)
        if tp.pooled:
            ng_csr = nodegraph.device_csr(dev)
            if ng_csr['n_src'] != M: raise ValueError('graph focus: every node must pass the mask')
            tp.y_pred = p.aggregate(ng_csr, tp.out_nodes, T, p.new(ng_csr['n_dst'], T))
        else:
            tp.y_pred = tp.out_nodes
        return tp

    def loss_and_grad(self, tp, y_pred, y, sample_weight):
        """(loss scalar tensor, d loss / d y_pred) with Keras semantics (SUM_OVER_BATCH_SIZE, sample weights)."""
        m, p = self.model, tp.p
        y = y.to(tp.dev, torch.float32).contiguous()
        sw = None if sample_weight is None else sample_weight.to(tp.dev, torch.float32).contiguous()
        R, T = y_pred.shape
        dpred, loss_rows = p.new(R, T), p.zeros(max(R, 1))
        kind = m.loss if isinstance(m.loss, str) else getattr(m.loss, '__name__', str(m.loss))
        if kind.lower() not in nat.LOSSES: raise ValueError(f'loss {kind!r} has no device gradient')
        p.loss_grad(nat.LOSSES[kind.lower()], y, y_pred.contiguous(), sw, dpred, loss_rows)
        if self.dp is not None:
            # SUM_OVER_BATCH_SIZE over the target rows of the merged batch: the kernel divided by this shard's row count
            R_all = self.dp.total_rows('targets', R)
            total = loss_rows[:R].sum().reshape(1)
            self.dp.all_reduce_sum(total)
            dpred.mul_(R / R_all)
            return total[0] / max(R_all, 1), dpred
        return loss_rows[:R].sum() / max(R, 1), dpred

    def pool_backward(self, tp, dpred):
        """d loss / d node-level outputs from d loss / d pooled graph outputs: NodeGraph . dpred (GNN.py:345 transposed)."""
        return tp.p.aggregate(_by_source(tp.nodegraph, tp.dev), dpred, tp.T, tp.p.new(tp.M, tp.T))

    # ---- backward sweep -------------------------------------------------------------------------------------------------
    def backward(self, tp, G_out_nodes, d_state_extra=None, want_label_grads=False, want_arc_label_grads=False):
        """Accumulates the parameter gradients into tp.gs / tp.go from `G_out_nodes` = d loss / d tp.out_nodes (M x T)
        and an optional extra gradient on the final state (N x S; LGNN feeds the next layer's label gradient here).
        `want_label_grads` (homogeneous models): also returns d loss / d nodes (N x L), the quantity an LGNN layer hands
        to the layer below through `update_graph` (reference LGNN.py:175-214).
        `want_arc_label_grads` (homogeneous models): leaves d loss / d arc labels (E x A) in `tp.d_arc_labels` — through the
        ArcNode scatter-add of every iteration's constant segment (GNN.py:254) and, for arc focus, the output network's
        arc-label segment (GNN.py:326): what an arc-focused LGNN layer with `get_output` hands to the layer below."""
        m, p = self.model, tp.p
        self.prim, self.drop_seed = p, tp.drop_seed                 # the recomputed calls draw the forward's Dropout masks
        self.go = tp.go
        N, S, L, d, M, k, focus = tp.N, tp.S, tp.L, tp.d, tp.M, tp.k, tp.focus
        gs, go, rows, rows_long, counts = tp.gs, tp.go, tp.rows, tp.rows_long, tp.counts
        G_state = p.zeros(N, S)                                     # dL / d states[k]
        if d_state_extra is not None: G_state.add_(d_state_extra)
        d_nodes = p.zeros(N, L) if want_label_grads else None
        if M > 0 and G_out_nodes is not None:
            lab = want_label_grads and tp.with_labels
            if focus == 'a':
                stride = 2 if tp.with_labels else 1
                dxs, dxd = p.new(M, S), p.new(M, S)
                req = [(0, dxs), (stride, dxd)]
                if lab:
                    dls, dld = p.new(M, L), p.new(M, L)
                    req += [(1, dls), (3, dld)]
                d_arc_o = None
                if want_arc_label_grads and tp.A > 0:
                    d_arc_o = p.new(M, tp.A); req.append((len(tp.osegs) - 1, d_arc_o))      # the arc-label segment is the last one
                self._mlp_backward(go, tp.osegs, tp.ohs, G_out_nodes, M, tp.ostats, req)
                p.scatter_add_rows(dxs, tp.isrc, G_state); p.scatter_add_rows(dxd, tp.idst, G_state)
                if lab: p.scatter_add_rows(dls, tp.isrc, d_nodes); p.scatter_add_rows(dld, tp.idst, d_nodes)
                if d_arc_o is not None: tp._d_arc_out = d_arc_o
            else:
                dxo = p.new(M, S)
                req = [(0, dxo)]
                if lab:
                    dlo = p.new(M, L); req.append((1, dlo))
                self._mlp_backward(go, tp.osegs, tp.ohs, G_out_nodes, M, tp.ostats, req)
                p.scatter_add_rows(dxo, tp.out_index, G_state)
                if lab: p.scatter_add_rows(dlo, tp.out_index, d_nodes)

        dx_s, dx_a = p.zeros(N, S), p.zeros(N, S)
        d_aggn = dl_t = da_t_ = d_acomp = None
        if want_label_grads and d > 0 and not tp.composite:
            d_aggn, dl_t, da_t_ = p.zeros(N, L), p.new(N, L), p.new(N, L)
        if want_label_grads and tp.composite:
            d_acomp = p.zeros(N, max(sum(tp.dims) + tp.A, 1))
        arc_grads = want_arc_label_grads and not tp.composite and tp.A > 0
        if arc_grads: d_aarcs, da_arcs_t = p.zeros(N, tp.A), p.new(N, tp.A)
        for t in range(k - 1, -1, -1):
            p.aggregate(tp.adj, tp.states[t], S, tp.agg)
            for ty in range(tp.T_types):
                if tp.all_counts[ty] == 0: continue
                segs, i_state, i_agg, i_lab, i_alab = tp.state_segs(t, ty)
                if counts[ty] == 0:                                  # data parallel: the type's rows live on other ranks (zeros into the sums)
                    self._mlp_backward(gs[ty], segs, None, None, 0, tp.stats_t[t][ty], None)
                    continue
                hs, _ = self._mlp_forward(gs[ty], segs, counts[ty], stats=tp.stats_t[t][ty], call=t)
                if rows[ty] is None:
                    req = [(i_state, dx_s), (i_agg, dx_a)]
                    if want_label_grads and d > 0: req += [(i_lab, dl_t), (i_alab, da_t_)]
                    if arc_grads: req.append((len(segs) - 1, da_arcs_t))             # the aggregated-arc segment is the last one
                    self._mlp_backward(gs[ty], segs, hs, G_state, N, tp.stats_t[t][ty], req)
                    if want_label_grads and d > 0:
                        p.axpby(1.0, d_nodes, 1.0, dl_t, d_nodes); p.axpby(1.0, d_aggn, 1.0, da_t_, d_aggn)
                    if arc_grads: p.axpby(1.0, d_aarcs, 1.0, da_arcs_t, d_aarcs)
                else:
                    G_t = G_state.index_select(0, rows_long[ty])
                    ds_t, da_t = p.new(counts[ty], S), p.new(counts[ty], S)
                    req = [(i_state, ds_t), (i_agg, da_t)]
                    if want_label_grads:                             # composite: own labels + aggregated component
                        if i_lab is not None:
                            dlab = p.new(counts[ty], tp.dims[ty]); req.append((i_lab, dlab))
                        if i_alab is not None:
                            dac = p.new(counts[ty], segs[i_alab][0].shape[1]); req.append((i_alab, dac))
                    self._mlp_backward(gs[ty], segs, hs, G_t, counts[ty], tp.stats_t[t][ty], req)
                    dx_s.index_copy_(0, rows_long[ty], ds_t); dx_a.index_copy_(0, rows_long[ty], da_t)
                    if want_label_grads:
                        if i_lab is not None: d_nodes[:, :tp.dims[ty]].index_add_(0, rows_long[ty], dlab)
                        if i_alab is not None: d_acomp.index_add_(0, rows_long[ty], dac)
            p.aggregate(tp.adj_src, dx_a, S, G_state)               # Adj . d agg   (arcs walked by source)
            p.axpby(1.0, G_state, 1.0, dx_s, G_state)
        if want_arc_label_grads and not tp.composite:
            # d arc labels = ArcNode . d agg_arcs (the scatter-add of GNN.py:254 transposed: every arc reads its destination's
            # row, times its weight) + the output network's arc-label segment on the masked arcs (arc focus)
            E = tp.arc_labels.shape[0]
            d_arcs = p.zeros(E, max(tp.A, 1))[:, :tp.A]
            if arc_grads and k > 0:
                d_arcs = p.aggregate(_by_source(tp.arcnode, tp.dev), d_aarcs, tp.A, p.new(E, tp.A))
            extra = getattr(tp, '_d_arc_out', None)
            if extra is not None:
                d_arcs = d_arcs.contiguous()
                p.scatter_add_rows(extra, tp.out_index, d_arcs)
            tp.d_arc_labels = d_arcs
        if want_label_grads and tp.composite:
            col = 0
            for t_src, ca in enumerate(tp.cas):                     # through CA_t^T . nodes[:, :d_t] (CompositeGNN.py:251)
                dt = tp.dims[t_src]
                if dt > 0:
                    tmp = p.aggregate(_by_source(ca, tp.dev), d_acomp[:, col:col + dt], dt, p.new(N, dt))
                    d_nodes[:, :dt].add_(tmp)
                col += dt
            if d == 0: d_nodes.add_(G_state)                        # state_0 = nodes (CompositeGNN.py:258)
        elif want_label_grads:
            if d > 0:
                tmp = p.aggregate(tp.adj_src, d_aggn, L, p.new(N, L))   # through the label aggregate Adj^T . nodes (GNN.py:258)
                p.axpby(1.0, d_nodes, 1.0, tmp, d_nodes)
            else:
                d_nodes = G_state                                   # state_0 = nodes (GNN.py:259)
        return d_nodes

    def finish(self, tp, apply=True):
        """Zero gradients of untouched networks, add the weight-penalty gradients, 1/k on the state gradients when
        `average_st_grads` (GNN.py:295: the division covers the whole gradient, penalty included), optimizer update.
        Returns the regularization loss (0-dim tensor) or None."""
        m = self.model
        reg = None
        for g_ in tp.gs:
            if tp.k == 0 or not g_.touched:
                for g in g_.gradients(): g.zero_()
        if not tp.go.touched:
            for g in tp.go.gradients(): g.zero_()
        if self.dp is not None:
            # layers behind the first: dW_l = h_{l-1}^T dZ_l summed over the rows of every shard (the first layer's gradients were
            # formed from the all-reduced P, q and are global already)
            upper = [g for g_ in list(tp.gs) + [tp.go] for l in range(1, len(g_.dW)) for g in (g_.dW[l], g_.db[l])]
            if upper: self.dp.all_reduce_sum(*upper)
        for g_ in list(tp.gs) + [tp.go]:
            r = _regularize(g_)
            if r is not None: reg = r if reg is None else reg + r
        for g_ in tp.gs:
            if m.average_st_grads and tp.k > 0:
                for g in g_.gradients(): g.mul_(1.0 / tp.k)
        if apply:
            opt, gate = m._optimizer_obj(), getattr(tp, 'grads_ok', None)
            if gate and not isinstance(opt, (Adam, SGD)):
                # a foreign optimizer knows nothing of the validity word: read it here (one synchronisation) instead of handing it over
                if not self._read_word(getattr(tp, 'grads_ok_view', None)): return reg
                gate = None
            if gate: opt.apply_gradients(self.grads_and_vars(tp), gate=gate)
            else: opt.apply_gradients(self.grads_and_vars(tp))
        return reg

    # ---- validity of an in-library step's gradients (include/gnnloop.h, ABI 7) ---------------------------------------------------------
    # The persistent small-graph BACKWARD kernel waits at grid barriers with a bound (GNN_WAIT_MS); when a wait expires its gradients are
    # NaN - and the call has long returned.  The library therefore leaves a validity word on the device: the step's moving-average updates
    # and the optimizer launch are gated by it (a failed step changes nothing), and the host learns of it without a synchronisation of its
    # own: the NEXT `gnn_train_step` fetches the word at its one synchronisation (`prev_grads_ok_host`), `fit()` asks at the end of an
    # epoch (`resolve_pending`).  The batch of a failed step is then trained on the building blocks (no cross-workgroup waits), late by
    # one step, with a RuntimeWarning.
    _pending = None

    @staticmethod
    def _read_word(view):
        return True if view is None else bool(int(view.item()) != 0)

    def resolve_pending(self):
        """Ask whether the last in-library step's gradients were valid (synchronises); trains its batch on the building blocks if not."""
        pend, self._pending = self._pending, None
        if pend is not None and not self._read_word(pend['view']): self._recover_failed(pend)

    def _uncount_gated_step(self):
        """The optimizer launch of a failed step was gated off on the device but counted on the host (`opt.iterations`, Adam's bias
        correction): take it back BEFORE the next update is issued, so that update n is always computed with step = n."""
        opt = self.model._optimizer_obj()
        if hasattr(opt, 'iterations') and isinstance(opt.iterations, int) and opt.iterations > 0: opt.iterations -= 1

    def _recover_failed(self, pend, uncount=True):
        import warnings
        warnings.warn('the persistent backward kernel of a training step could not keep its workgroups resident (GPU shared with other '
                      'long-running work?): its gradients were discarded on the device - weights, optimizer slots and moving statistics '
                      'untouched - and its batch is trained on the general kernels now', RuntimeWarning, stacklevel=4)
        self.recovered_steps = getattr(self, 'recovered_steps', 0) + 1
        if uncount: self._uncount_gated_step()                  # (the gated launch did not count)
        x_list, y, sample_weight, state0, seed = pend['batch']
        self._train_step_general(x_list, y, sample_weight, state0, seed, True)

    @staticmethod
    def grads_and_vars(tp):
        grads = [g for g_ in tp.gs for g in g_.gradients()] + tp.go.gradients()
        variables = [v for g_ in tp.gs for v in g_.variables()] + tp.go.variables()
        return list(zip(grads, variables))

    # ---- one training step ------------------------------------------------------------------------------------------------
    # ---- the whole step inside the library (homogeneous models): gnn_train_step -------------------------------------------------
    use_native_step = True          # False: always the general path below (tests compare the two)
    use_tiles = True                # False: never hand the batch's diagonal blocks to gnn_train_step (tests compare the two forms)

    def _cached_grads(self, name, net, p):
        c = getattr(self, '_ng_' + name, None)
        w = net.weights
        if c is None or c.net is not net or len(c._weights) != len(w) or any(a is not b for a, b in zip(c._weights, w)):
            c = _NetGrads(net, p)
            c._weights = list(w)
            setattr(self, '_ng_' + name, c)
        c.touched = False
        return c

    def _native_step_applies(self, y):
        m = self.model
        if self.dp is not None: return False        # collectives sit between the iteration's launches: the building-block path
        if not self.use_native_step or y is None: return False
        nets = list(m.net_state) if isinstance(m.net_state, (list, tuple)) else [m.net_state]
        if isinstance(m.net_state, (list, tuple)) and m.max_iteration < 1: return False      # what the in-library composite step refuses (make_cplan): the general path below
        # Dropout layers behind Dense layers run inside the library (ABI 8); one in FRONT of the first Dense (position 0) keeps the path below
        for n_ in nets + [m.net_output]:
            if any(float(r) > 0 and int(q) == 0 for r, q in zip(n_.dropout_rate or [], n_.dropout_pos or [])): return False
        kind = m.loss if isinstance(m.loss, str) else getattr(m.loss, '__name__', str(m.loss))
        return str(kind).lower() in nat.LOSSES

    def _native_forward_applies(self):
        """May `Loop(..., training=True)` run as ONE `gnn_train_step(forward_only)` call?  (homogeneous models, no Dropout in front of a first
        Dense, no data parallelism - what the in-library step covers, minus everything about the loss.)"""
        m = self.model
        if self.dp is not None or not self.use_native_step or isinstance(m.net_state, (list, tuple)) or m.max_iteration < 1: return False
        for n_ in (m.net_state, m.net_output):
            if any(float(r) > 0 and int(q) == 0 for r, q in zip(n_.dropout_rate or [], n_.dropout_pos or [])): return False
        return True

    def forward_native(self, x_list, state0=None, seed=None, node_level=False):
        """The training-mode forward in one library call (include/gnnloop.h ABI 9, `forward_only`): (k, state, output rows) - the same
        arithmetic as `forward()` on the building blocks, no tape kept.  `node_level`: a graph-focused model stops at its per-node outputs."""
        r = self._train_step_native(x_list, None, None, state0, seed, False, forward_only=True, node_level=node_level)
        return r['k'], r['state'], r['y_pred']

    def _train_step_native(self, x_list, y, sample_weight, state0, seed, apply, forward_only=False, node_level=False):
        """One `gnn_train_step` call: training-mode forward, loss, BPTT; the tape and every scratch buffer live in one cached
        device allocation. Same results as the general path (same kernels for the arithmetic), ~2.5x fewer launches and no
        Python between them.  `forward_only`: the forward alone (no targets, no gradients; `Loop(..., training=True)`)."""
        from types import SimpleNamespace
        from .GNN import _squeeze_last, _arc_endpoints
        m = self.model
        composite = isinstance(m.net_state, (list, tuple))
        nets_s = list(m.net_state) if composite else [m.net_state]
        inputs = m.process_inputs(x_list)
        if composite:       # CompositeGNN.py:275-304: one state network per node type (csrc/train_composite.hpp)
            nodes, arcs, dim_node_label, type_mask, set_mask, output_mask, cas, adjacency, arcnode, nodegraph = inputs
        else:
            nodes, arcs, dim_node_label, set_mask, output_mask, adjacency, arcnode, nodegraph = inputs
        nat.require_device(nodes, 'nodes'); nat.require_device(arcs, 'arcs')
        dev = nodes.device
        self.prim = p = _Prim(dev)
        nodes = nodes.to(torch.float32).contiguous(); arcs = arcs.to(torch.float32).contiguous()
        N, L = nodes.shape
        d = m.state_vect_dim
        S = d if d > 0 else L
        focus = 'n' if (forward_only and node_level and m._focus == 'g') else m._focus
        for n_ in nets_s: n_.to(dev)
        m.net_output.to(dev)
        if forward_only: self.resolve_pending()                  # (the tape is shared with the steps: nothing of theirs may be pending on it)
        # gnn_train_step OVERWRITES every gradient buffer: the holders of the previous step are reused as they are (eight zero-fill
        # launches and their allocations per step otherwise; the optimizer's pointer tables stay valid too)
        gs_all = [] if forward_only else [self._cached_grads(f'state{i}' if composite else 'state', n_, p) for i, n_ in enumerate(nets_s)]
        if not forward_only:
            gs, go = gs_all[0], self._cached_grads('output', m.net_output, p)
            self.gs, self.go = (gs_all if composite else gs), go
        out_index = m._out_index(_squeeze_last(set_mask).to(dev), _squeeze_last(output_mask).to(dev))
        adj, an = adjacency.device_csr(dev), arcnode.device_csr(dev)
        keep = [nodes, arcs, adj, an, out_index]
        ta = nat.TrainArgs()
        a = ta.loop
        a.abi_version, a.composite = nat.GNN_ABI_VERSION, 0
        a.n_nodes, a.n_arcs, a.dim_node_label, a.dim_arc_label = N, arcs.shape[0], L, arcs.shape[1] - 2
        a.nodes, a.ld_nodes = nat.ptr(nodes), L
        a.arc_labels, a.ld_arcs = C.c_void_p(arcs.data_ptr() + 8), arcs.shape[1]
        a.adjacency, a.arcnode = nat.make_csr(adj), nat.make_csr(an)
        a.n_types = 1
        if composite:
            from .GNN import _squeeze_last as _sq
            dims = [int(v) for v in (dim_node_label.reshape(-1).tolist() if isinstance(dim_node_label, torch.Tensor)
                                     else np.asarray(dim_node_label).reshape(-1))]
            if len(dims) != len(nets_s): raise ValueError(f'{len(dims)} node types but {len(nets_s)} state networks')
            type_nodes, offsets = m._type_lists(_sq(type_mask).to(dev))
            ca_csr = [c_.device_csr(dev) for c_ in cas]
            keep += [type_nodes, ca_csr]
            a.composite, a.n_types = 1, len(dims)
            a.type_nodes = nat.ptr(type_nodes)
            for t_, dt in enumerate(dims):
                a.type_dim_label[t_] = dt
                a.type_offsets[t_] = int(offsets[t_])
                a.composite_adjacency[t_] = nat.make_csr(ca_csr[t_])
                a.net_state[t_] = nets_s[t_].native()
            a.type_offsets[len(dims)] = int(offsets[len(dims)])
        else:
            a.net_state[0] = m.net_state.native()
        a.net_output = m.net_output.native()
        a.state_dim, a.max_iteration, a.state_threshold = d, m.max_iteration, float(m.state_threshold)
        if d > 0:
            if state0 is None:
                gen = None
                if seed is not None:
                    gen = torch.Generator(device=dev); gen.manual_seed(int(seed))
                state0 = torch.empty((N, d), device=dev, dtype=torch.float32).normal_(0.0, 0.1, generator=gen)      # tf.random.normal(stddev=0.1), one launch
            state0 = state0.to(dev, torch.float32).contiguous()
            if tuple(state0.shape) != (N, d): raise ValueError('state0 must be (n_nodes, state_vect_dim)')
            a.state0 = nat.ptr(state0); keep.append(state0)
        a.focus = nat.FOCUS[focus]
        a.n_out, a.out_index = len(out_index), nat.ptr(out_index)
        if not forward_only: ta.adjacency_by_source = nat.make_csr(_by_source(adjacency, dev))
        if focus == 'a':
            es, ed = _arc_endpoints(adjacency, dev)
            a.arc_src, a.arc_dst = nat.ptr(es), nat.ptr(ed); keep += [es, ed]
        n_rows = len(out_index)
        if focus == 'g':
            ng = nodegraph.device_csr(dev)
            a.nodegraph = nat.make_csr(ng); keep.append(ng)
            if not forward_only: ta.nodegraph_by_source = nat.make_csr(_by_source(nodegraph, dev))
            n_rows = ng['n_dst']
        a.stream = p.stream()
        ta.forward_only = int(bool(forward_only))
        if not forward_only:
            yd = y.to(dev, torch.float32).contiguous()
            if yd.shape[0] != n_rows: raise ValueError(f'targets have {yd.shape[0]} rows, the model outputs {n_rows}')
            sw = None if sample_weight is None else sample_weight.to(dev, torch.float32).contiguous()
            ta.targets, ta.sample_weight = nat.ptr(yd), nat.ptr(sw)
            kind = m.loss if isinstance(m.loss, str) else getattr(m.loss, '__name__', str(m.loss))
            ta.loss_kind = nat.LOSSES[str(kind).lower()]
        ta.average_st_grads = 0          # finish() divides by k AFTER adding the weight penalties, like the reference (GNN.py:295)
        ta.bn_momentum = BN_MOMENTUM
        # Dropout / AlphaDropout layers (reference MLP.py:60-66): their masks derive from the step's seed exactly as on the building-block path
        m._dropout_step = getattr(m, '_dropout_step', 0) + 1
        self.drop_seed = _mix32(0x5EED, int(seed)) if seed is not None else _mix32(id(m) & 0xFFFFFFFF, m._dropout_step)
        ta.drop_seed = self.drop_seed
        drop_nets = [self._cached_grads('state', m.net_state, p), self._cached_grads('output', m.net_output, p)] if forward_only else None
        if forward_only: gs_all, go = [drop_nets[0]], drop_nets[1]           # (the holders know the networks' Dropout layers; their buffers stay untouched)
        for i, g_ in enumerate(gs_all): g_.net_id = i
        go.net_id = 1000
        for spec, g_ in [(ta.drop_state[i], g_) for i, g_ in enumerate(gs_all)] + [(ta.drop_output, go)]:
            layers = [(q, r, idx) for q in sorted(g_.drop) for r, idx in g_.drop[q]]
            if len(layers) > nat.GNN_MAX_DROPOUT: raise ValueError(f'more than {nat.GNN_MAX_DROPOUT} Dropout layers in one network')
            spec.n, spec.alpha, spec.net_id = len(layers), int(bool(g_.alpha)), int(g_.net_id)
            for j, (q, r, idx) in enumerate(layers): spec.pos[j], spec.rate[j], spec.index[j] = int(q), float(r), int(idx)
        holders = [] if forward_only else ([(ta.grad_state_types[i], g_) for i, g_ in enumerate(gs_all)] if composite else [(ta.grad_state, gs)]) + [(ta.grad_output, go)]
        for g_, ng_ in holders:
            if ng_.bn: g_.dgamma, g_.dbeta = nat.ptr(ng_.dgamma), nat.ptr(ng_.dbeta)
            for l in range(len(ng_.W)): g_.dkernel[l], g_.dbias[l] = ng_.dW[l].data_ptr(), ng_.db[l].data_ptr()
        T = m.net_output.units[-1]
        y_pred, state = p.new(n_rows, T), p.new(N, S)
        loss = p.new(1)
        k_host = C.c_int32(0)
        ta.y_pred, ta.state, ta.loss, ta.k_host = nat.ptr(y_pred), nat.ptr(state), nat.ptr(loss), C.pointer(k_host)
        tiles = adjacency.tiles(64) if (self.use_tiles and not composite) else None      # a merged batch: graphs packed into tiles of <= 64 nodes
        if tiles is not None and len(tiles) - 1 <= 256:
            ta.tile_node_begin, ta.n_tiles = tiles.ctypes.data, len(tiles) - 1
        nbytes = nat.lib().gnn_train_workspace_bytes(C.byref(ta))
        if nbytes == 0: nat.check(1)
        tape = getattr(self, '_tape', None)
        if tape is None or tape.numel() < nbytes + 256 or tape.device != dev:
            self.resolve_pending()                           # (the previous step's validity word lives in the tape that is about to go)
            tape = self._tape = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
        base = tape.data_ptr()
        aligned = (base + 255) & ~255
        ta.tape, ta.tape_bytes = C.c_void_p(aligned), tape.numel() - (aligned - base)
        if forward_only:
            nat.check(nat.lib().gnn_train_step(C.byref(ta)))
            return {'k': int(k_host.value), 'y_pred': y_pred, 'state': state}
        # the validity word of this step's gradients (first word of the tape) and - for free, at the call's one synchronisation - the one
        # the previous step left there
        ok_ptr, prev_ok = C.c_void_p(0), C.c_int32(1)
        ta.grads_ok_dev = C.pointer(ok_ptr)
        pend, self._pending = self._pending, None
        fetched_by_call = pend is not None and pend['tape'] is tape
        if fetched_by_call:
            prev_ok.value = -1                                   # sentinel: the library has not fetched the previous step's word (yet)
            ta.prev_grads_ok_host = C.pointer(prev_ok)
        elif pend is not None and not self._read_word(pend['view']): prev_ok.value = 0
        try:
            nat.check(nat.lib().gnn_train_step(C.byref(ta)))
        except nat.NativeError:
            # The call fetches the previous step's validity word and then RESETS it on the tape for its own step: once it has got that far
            # (`prev_ok` is 0 / 1 - a failing call synchronises before it returns), the word on the tape is this call's zero and says nothing
            # about the previous step; only a call that failed in front of its first launch (the sentinel is still there) left it untouched.
            if pend is not None:
                prev_valid = bool(prev_ok.value) if (fetched_by_call and prev_ok.value >= 0) else (self._read_word(pend['view']) if fetched_by_call else prev_ok.value != 0)
                if not prev_valid: self._recover_failed(pend)
            raise
        for g_ in gs_all: g_.touched = k_host.value > 0          # (a type without nodes: the library zero-fills its gradients)
        go.touched = len(out_index) > 0
        view = tape[aligned - base:aligned - base + 4].view(torch.int32)
        tp = SimpleNamespace(gs=gs_all, go=go, k=int(k_host.value), y_pred=y_pred, state=state, grads_ok=ok_ptr.value, grads_ok_view=view)
        res = {'k': tp.k, 'y_pred': y_pred, 'state': state, 'loss': loss[0]}
        failed_prev = pend is not None and prev_ok.value == 0
        if failed_prev: self._uncount_gated_step()              # before THIS step's update is issued: it is update n, not n + 1
        reg = self.finish(tp, apply)
        if reg is not None: res['loss'] = res['loss'] + reg
        if apply and ok_ptr.value: self._pending = {'tape': tape, 'view': view, 'batch': (x_list, y, sample_weight, state0, seed)}
        res['grads_ok'] = view                                   # device int32[1]: 1 when the gradients / the update of this step are valid
        if failed_prev: self._recover_failed(pend, uncount=False)      # the PREVIOUS step's batch, one step late
        return res

    def train_step(self, x_list, y, sample_weight, state0=None, seed=None, apply=True):
        """Returns dict(loss=..., k=..., y_pred=tensor). Gradients stay in self.gs (list for composite models) / self.go;
        `apply` runs the optimizer. Homogeneous models are the one-type case (reference GNN.py:277-306) and run the whole step
        inside the library (`gnn_train_step`); composite models (CompositeGNN.py:275-304) run one state network per node type on
        that type's rows through the general path below."""
        m = self.model
        if self._native_step_applies(y):
            try:
                return self._train_step_native(x_list, y, sample_weight, state0, seed, apply)
            except nat.NativeError as e:
                # the persistent forward kernel's grid barrier expired (GPU shared with other long-running work: not all of its
                # workgroups resident within GNN_WAIT_MS).  The in-library step checks that right behind the forward loop - before the
                # moving statistics, the gradients or the weights are touched - so the step is simply run again on the building
                # blocks below, which have no cross-workgroup waits.
                # a configuration the in-library step does not cover (its plan says so before anything runs) simply takes the general path
                not_covered = any(t in str(e) for t in ('train through the building blocks', 'requires max_iteration', 'empty graph', 'homogeneous models only'))
                if 'grid barrier' not in str(e) and 'cannot be resident' not in str(e) and not not_covered: raise
                if not_covered:
                    self.resolve_pending()
                    return self._train_step_general(x_list, y, sample_weight, state0, seed, apply)
                import warnings
                warnings.warn('the persistent training kernels could not keep their workgroups resident (GPU shared with other '
                              'long-running work?): this step runs on the general kernels', RuntimeWarning, stacklevel=3)
                self.recovered_steps = getattr(self, 'recovered_steps', 0) + 1
        self.resolve_pending()
        return self._train_step_general(x_list, y, sample_weight, state0, seed, apply)

    def _train_step_general(self, x_list, y, sample_weight, state0, seed, apply):
        """The step on the building blocks, orchestrated from here (no cross-workgroup waits anywhere)."""
        m = self.model
        tp = self.forward(x_list, state0=state0, seed=seed)
        res = {'k': tp.k, 'y_pred': tp.y_pred, 'state': tp.state}
        if y is None:
            if m.loss: raise TypeError('Target data is missing. Your model was compiled with `loss` '
                                       'argument and so expects targets to be passed in `fit()`.')
            return res
        res['loss'], dpred = self.loss_and_grad(tp, tp.y_pred, y, sample_weight)
        G_out = self.pool_backward(tp, dpred) if tp.pooled else dpred
        self.backward(tp, G_out)
        reg = self.finish(tp, apply)
        if reg is not None: res['loss'] = res['loss'] + reg          # compiled_loss(..., regularization_losses=self.losses)
        return res


def mlp_training_call(net: Sequential, segs, M, seed=None, net_id=0, call=0):
    """Keras TRAINING-mode call of one reference MLP (`net(x, training=True)`, reference MLP.py:60-78 layers in training mode): the
    input is the virtual concatenation `segs` = [(matrix view [*, w], row index or None), ...] of M rows; BatchNormalization
    normalises with the batch statistics of THIS call and moves its moving averages once (momentum 0.99), Dropout / AlphaDropout
    draw a fresh mask (`seed` makes it reproducible).  What the standalone `convergence(..., training=True)` and
    `Sequential.__call__(x, training=True)` run; `Loop(training=True)` / `train_step` use the same primitives with their tape."""
    dev = segs[0][0].device
    t = LoopTrainer.__new__(LoopTrainer)
    t.model, t.dp, t.go = None, None, None
    t.prim = LoopTrainer.prim_cls(dev)
    net.to(dev)
    net._training_calls = getattr(net, '_training_calls', 0) + 1
    t.drop_seed = _mix32(0x5EED, int(seed)) if seed is not None else _mix32(id(net) & 0xFFFFFFFF, net._training_calls)
    ng = _NetGrads.forward_only(net)
    ng.net_id = net_id
    hs, st = t._mlp_forward(ng, segs, M, call=call)
    if ng.bn and M > 0:
        mm, mv = ng.moving
        mm.mul_(BN_MOMENTUM).add_(st[0] * (1 - BN_MOMENTUM)); mv.mul_(BN_MOMENTUM).add_(st[1] * (1 - BN_MOMENTUM))
    return hs.out


def _regularize(ng: _NetGrads):
    """Adds d(penalty)/dw of every regularized Dense variable of one network to its gradient buffers and returns the penalty
    `sum_l l1 * sum|w| + l2 * sum(w^2)` (each variable once per step, however many times the loop applied the network)."""
    net, total = ng.net, None
    for l in range(len(net.units)):
        for reg, w, g in ((net.kernel_regularizer[l], ng.W[l], ng.dW[l]), (net.bias_regularizer[l], ng.b[l], ng.db[l])):
            if reg is None or (reg.l1 == 0.0 and reg.l2 == 0.0): continue
            pen = reg.l1 * w.abs().sum() + reg.l2 * (w * w).sum()
            g.add_(reg.l1 * torch.sign(w) + (2.0 * reg.l2) * w)
            total = pen if total is None else total + pen
    return total


def _by_source(matrix: SparseMatrix, device):
    """CSR of A itself (arcs grouped by SOURCE): the operator of the transposed aggregate, out[i] = Σ_{e: src=i} w_e X[dst_e]."""
    key = ('by_source', str(device))
    if key not in matrix._dev and hasattr(matrix, 'by_source'):
        return matrix.by_source(device)            # assembled on the device (gnnkeras_amd/device_batch.py)
    if key not in matrix._dev:
        c = CSRByDestination.from_coo(matrix.indices[:, 1], matrix.indices[:, 0], matrix.values,
                                      (matrix.dense_shape[1], matrix.dense_shape[0]))
        up = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)
        matrix._dev[key] = dict(rowptr=up(c.rowptr), src=up(c.src), w=up(c.w), row_scale=up(c.row_scale), n_src=c.n_src,
                                n_dst=c.n_dst, nnz=c.nnz)
    return matrix._dev[key]
