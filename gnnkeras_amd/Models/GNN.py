"""Homogeneous recurrent GNN (Scarselli et al. 2009) on MI355X: node-, arc- and graph-focused models.

Host-side mirror of the reference's `GNN/Models/GNN.py` — same constructor, `compile`, `call`, `process_inputs`,
`condition`, `convergence`, `apply_filters`, `Loop`, `evaluate` / `predict` names and argument meaning — with the
whole `Loop` (reference `GNN.py:245-274`) executed by `gnn_loop_forward` of libgnnloop.so: hand-written HIP kernels,
`max_iteration` gated launches, no host synchronisation per iteration; `k` comes back as a device scalar.
The Python `while` of the reference (`tf.while_loop` in eager mode, `GNN.py:265`) does not exist here.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .. import _native as nat
from .. import ops
from ..sparse import SparseMatrix
from .MLP import Sequential


# ----------------------------------------------------------------------------------------------------------------------
# losses / metrics of the Keras-style evaluate(): tiny torch reductions on the device, outside the hot path
# ----------------------------------------------------------------------------------------------------------------------
def _loss_fn(loss):
    if callable(loss): return loss
    name = str(loss).lower()
    eps = 1e-7
    if name in ('categorical_crossentropy', 'cce'):
        def f(y, p):
            p = p / p.sum(-1, keepdim=True)
            return -(y * torch.log(p.clamp(eps, 1 - eps))).sum(-1)
        return f
    if name in ('binary_crossentropy', 'bce'):
        def f(y, p):
            p = p.clamp(eps, 1 - eps)
            return -(y * torch.log(p) + (1 - y) * torch.log(1 - p)).mean(-1)
        return f
    if name in ('mse', 'mean_squared_error'): return lambda y, p: ((y - p) ** 2).mean(-1)
    if name in ('mae', 'mean_absolute_error'): return lambda y, p: (y - p).abs().mean(-1)
    raise ValueError(f'unknown loss {loss!r}')


def _metric_fn(metric, n_targets):
    if callable(metric): return getattr(metric, '__name__', 'metric'), metric
    name = str(metric).lower()
    if name in ('accuracy', 'acc'):
        if n_targets > 1: return 'accuracy', lambda y, p: (y.argmax(-1) == p.argmax(-1)).float()
        return 'accuracy', lambda y, p: ((p > 0.5).float() == y).float().mean(-1)
    if name in ('categorical_accuracy',): return name, lambda y, p: (y.argmax(-1) == p.argmax(-1)).float()
    if name in ('binary_accuracy',): return name, lambda y, p: ((p > 0.5).float() == y).float().mean(-1)
    if name in ('mse', 'mae'): return name, _loss_fn(name)
    raise ValueError(f'unknown metric {metric!r}')


class History(dict):
    """What `fit` returns: {'loss': [per epoch], 'val_loss': [...], metrics...} - a dict (round-1/2 callers index it
    directly) that also answers Keras' `History.history` / `History.epoch`."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.epoch = []

    @property
    def history(self):
        return self


class _Launch(list):
    """One entry of a group plan: the batches a launch merges.  `resident`: every batch runs as one workgroup with its state
    in the LDS of one CU (no grid barrier), so the launch may overlap any other."""

    def __init__(self, batches, resident: bool = False, parts=None):
        super().__init__(batches)
        self.resident = resident
        self.parts = parts or {}      # batch -> node counts of the groups it was cut into (along graph boundaries); absent: one group

    def groups_and_sets(self, sizes):
        """(group_node_begin, group_set_begin | None) of the merged launch: `sizes[b]` = nodes of batch b."""
        begin, sets = [0], [0]
        for b in self:
            for n in self.parts.get(b, [sizes[b]]): begin.append(begin[-1] + int(n))
            sets.append(len(begin) - 1)
        return begin, (sets if self.parts else None)


class _LoopModel:
    """What the reference gets from `tf.keras.Model`: compile / evaluate / predict / fit plumbing around `call`."""

    def _engine_init(self):
        self.optimizer, self.loss, self.metrics_spec = None, None, []
        self.average_st_grads = None
        self._mask_cache = {}
        self._streams = None
        self.inference_streams = 8                     # side streams predict() / evaluate() spread their batches over

    def compile(self, *args, average_st_grads=False, **kwargs):
        """`compile(optimizer, loss, metrics=[...], average_st_grads=False)`; `run_eagerly` is accepted and irrelevant
        (reference forces it, GNN.py:157)."""
        names = ['optimizer', 'loss', 'metrics']
        for n, v in zip(names, args): kwargs.setdefault(n, v)
        self.optimizer = kwargs.get('optimizer', None)
        self.loss = kwargs.get('loss', None)
        self.metrics_spec = list(kwargs.get('metrics', None) or [])
        self.average_st_grads = average_st_grads

    def __call__(self, inputs, training: bool = False, mask=None):
        return self.call(inputs, training=training, mask=mask)

    def predict(self, sequencer, verbose=0, callbacks=None, **kwargs):
        """Outputs of every batch, concatenated (Keras `predict` semantics)."""
        self._check_kwargs('predict', kwargs)
        if len(sequencer) == 0: return np.zeros((0, 0), np.float32)
        dev = self._batch_device(sequencer[0][0])
        outs = self._with_recovery(lambda: [o for _, o in self._forward_batches(sequencer, dev)], dev)
        return torch.cat(outs, dim=0).cpu().numpy()

    def evaluate(self, sequencer, return_dict: bool = False, verbose=0, callbacks=None, **kwargs):
        """Loss and metrics over the sequencer with Keras `evaluate` semantics: the loss is sum(loss_i * weight_i) / number of
        samples (SUM_OVER_BATCH_SIZE per batch, batches averaged by their size), the metrics are weighted means
        sum(metric_i * weight_i) / sum(weight_i)."""
        self._check_kwargs('evaluate', kwargs)
        if self.loss is None: raise RuntimeError('compile() the model with a loss before evaluate()')
        lossf = _loss_fn(self.loss)
        if len(sequencer) == 0: raise ValueError('evaluate() needs at least one batch')
        dev = self._batch_device(sequencer[0][0])
        # every forward first (grouped launches / side streams), then ONE loss / metric evaluation over all samples: the sums
        # Keras accumulates batch by batch are the same sums, and a data set of small batches costs a handful of launches
        # instead of a dozen per batch
        preds = self._with_recovery(lambda: [p_ for _, p_ in self._forward_batches(sequencer, dev)], dev)
        p = torch.cat(preds, dim=0) if len(preds) > 1 else preds[0]
        # targets / sample weights of the whole sequencer, concatenated once per set of batches (a few sequencers per model:
        # training, validation, test); keyed by the batch list the sequencer rebuilds whenever its batches change
        owner = getattr(sequencer, 'graph_tensors', None)
        cache = self.__dict__.setdefault('_eval_targets', {})
        hit = cache.get(id(owner)) if owner is not None else None
        if hit is None or hit[0] is not owner or hit[1] != len(sequencer):
            ys, sws = zip(*[(sequencer[i][1], sequencer[i][2]) for i in range(len(sequencer))])
            hit = (owner, len(sequencer), torch.cat([y_.to(p.device) for y_ in ys], dim=0), torch.cat([w_.to(p.device) for w_ in sws], dim=0))
            if owner is not None:
                if len(cache) >= 4: cache.clear()
                cache[id(owner)] = hit
        y, sw = hit[2], hit[3]
        tot_loss = (lossf(y, p) * sw).sum()
        tot_w = float(sw.shape[0])
        cnt = sw.sum()
        mets = [(n, f, (f(y, p) * sw).sum()) for n, f in (_metric_fn(m, y.shape[-1]) for m in self.metrics_spec)]
        res = {'loss': float(tot_loss / tot_w)}
        for n, f, acc in mets: res[n] = float(acc / cnt)
        return res if return_dict else [res['loss']] + [res[n] for n, _, _ in mets]

    def _optimizer_obj(self):
        from .training import get_optimizer
        if getattr(self, '_opt_obj', None) is None or self._opt_src is not self.optimizer:
            self._opt_obj, self._opt_src = get_optimizer(self.optimizer), self.optimizer
        return self._opt_obj

    def train_step(self, data, *, state0=None, seed=None):
        """One optimisation step on one batch (reference GNN.py:277-306): training-mode forward, loss, BPTT through the
        k executed iterations, optional 1/k on the state gradients, optimizer update. Returns {'loss': ..., metrics...}."""
        from .training import LoopTrainer
        if self.loss is None: raise RuntimeError('compile() the model with a loss before fit() / train_step()')
        if getattr(self, '_trainer', None) is None: self._trainer = LoopTrainer(self)
        x, y, sample_weight = data
        res = self._trainer.train_step(x, y, sample_weight, state0=state0, seed=seed)
        out = {'loss': res['loss'], 'k': res['k']}
        if y is not None:
            yd = y.to(res['y_pred'].device)
            sw = torch.ones(yd.shape[0], device=yd.device) if sample_weight is None else sample_weight.to(yd.device)
            for m in self.metrics_spec:
                n, f = _metric_fn(m, yd.shape[-1])
                out[n] = (f(yd, res['y_pred']) * sw).sum() / sw.sum()
        return out

    # Keras arguments of fit / evaluate / predict that have no meaning for an eager single-process loop: accepted, ignored
    # (`shuffle`: Keras shuffles the ORDER of a Sequence's batches; the sequencers here re-draw the batches themselves at every
    # `on_epoch_end`, reference GraphSequencers.py:123-127, and are walked in order)
    _KERAS_NOOP_KWARGS = frozenset(('workers', 'use_multiprocessing', 'max_queue_size', 'batch_size', 'validation_batch_size', 'shuffle'))
    # Keras arguments that change the loss or the amount of training and are not implemented: accepted at their Keras default (no
    # effect), REFUSED otherwise - never silently ignored
    _KERAS_UNSUPPORTED_KWARGS = {'class_weight': None, 'sample_weight': None, 'validation_freq': 1, 'steps': None, 'steps_per_epoch': None,
                                 'validation_steps': None}

    @classmethod
    def _check_kwargs(cls, where, kwargs, allowed=()):
        """Unknown keyword arguments raise (Keras would, too) instead of vanishing."""
        refused = [k for k, v in kwargs.items() if k in cls._KERAS_UNSUPPORTED_KWARGS and not (v is None or (k == 'validation_freq' and isinstance(v, int) and v == 1))]
        if refused: raise NotImplementedError(f'{where}(): {refused} would change the loss / the amount of work and are not implemented here')
        bad = [k for k in kwargs if k not in cls._KERAS_NOOP_KWARGS and k not in cls._KERAS_UNSUPPORTED_KWARGS and k not in allowed]
        if bad: raise TypeError(f'{where}() got unexpected keyword argument(s) {bad}')

    def fit(self, sequencer, epochs: int = 1, validation_data=None, verbose: int = 1, callbacks=None, initial_epoch: int = 0, **kwargs):
        """Keras-style training loop over a sequencer: `train_step` per batch, `on_epoch_end` (reshuffle + re-merge,
        reference GraphSequencers.py:123-127) per epoch, optional validation with `evaluate`.  Returns a `History` (a dict of
        per-epoch lists that also has Keras' `.history` / `.epoch` attributes).

        `callbacks`: objects with any of `on_train_begin(logs)`, `on_epoch_begin(epoch, logs)`, `on_epoch_end(epoch, logs)`,
        `on_train_end(logs)` (the Keras Callback method names; `set_model(model)` is called when present).  A callback that sets
        `model.stop_training = True` ends the loop after the current epoch, as in Keras."""
        self._check_kwargs('fit', kwargs)
        cbs = list(callbacks or [])
        for cb in cbs:
            if hasattr(cb, 'set_model'): cb.set_model(self)
        def emit(name, *args):
            for cb in cbs:
                f = getattr(cb, name, None)
                if f is not None: f(*args)
        history = History()
        self.stop_training = False
        emit('on_train_begin', {})
        logs = {}
        for epoch in range(int(initial_epoch), epochs):
            emit('on_epoch_begin', epoch, {})
            vals, ws = {}, []
            for i in range(len(sequencer)):
                data = sequencer[i]
                r = self.train_step(data)
                ws.append(float(data[1].shape[0]) if data[1] is not None else 1.0)
                for key, val in r.items():
                    if key != 'k': vals.setdefault(key, []).append(val)        # device scalars stay on the device: no sync per step and value
            if getattr(self, '_trainer', None) is not None: self._trainer.resolve_pending()      # (the last step's validity word: training.py)
            wsum = float(sum(ws))
            logs = {}
            for key, lst in vals.items():                          # one transfer per logged quantity and epoch; the weighted mean in float64
                if all(isinstance(v, torch.Tensor) for v in lst):
                    v64 = torch.stack([v.detach().reshape(()) for v in lst]).to('cpu', torch.float64).numpy()
                else:
                    v64 = np.array([float(v) for v in lst], dtype=np.float64)
                logs[key] = float((v64 * np.asarray(ws[:len(lst)], dtype=np.float64)).sum() / max(wsum, 1.0))
            if validation_data is not None:
                logs.update({'val_' + key: val for key, val in self.evaluate(validation_data, return_dict=True).items()})
            for key, val in logs.items(): history.setdefault(key, []).append(val)
            history.epoch.append(epoch)
            if verbose:
                print(f'Epoch {epoch + 1}/{epochs} - ' + ' - '.join(f'{k_}: {v:.4f}' for k_, v in logs.items()))
            emit('on_epoch_end', epoch, logs)
            if hasattr(sequencer, 'on_epoch_end'): sequencer.on_epoch_end()
            if self.stop_training: break
        emit('on_train_end', logs)
        self.history = history
        return history

    _K_ERROR = ('a bounded in-launch wait of the loop kernels expired (persistent kernel: not all workgroups resident - GPU '
                'shared with other long-running work?; wave-specialised kernel: a lost LDS hand-off): state and output are '
                'invalid. Set model.native_flags = FLAG_FUSED_GEN2 or lower inference_streams')

    def _with_recovery(self, forwards, device=None):
        """`forwards()` (every forward of a predict() / evaluate() walk) with the kernels' in-launch waits checked where the host
        synchronises anyway.  The whole-loop kernels need all their workgroups resident at once and wait for each other with a
        bound (GNN_WAIT_MS); on a GPU shared with other long-running work such a wait can expire - reported as k < 0, results
        invalid.  The walk is then REPEATED on the kernels that have no cross-workgroup waits: one launch per iteration
        (`FLAG_FUSED_GEN2`), batch by batch, on the caller's stream - slower, always completes - with a `RuntimeWarning`.  A second
        failure (it would be a lost hand-off inside a workgroup: a bug, not contention) raises `NativeError`."""
        self._k_seen = []
        on_gpu = device is not None and torch.device(device).type == 'cuda'
        dev_rng = torch.cuda.get_rng_state(device) if on_gpu else None       # (host-side: seed + offset, no synchronisation)
        res = forwards()
        try:
            self._check_k()
            return res
        except nat.NativeError:
            pass
        if dev_rng is not None: torch.cuda.set_rng_state(dev_rng, device)      # the repeated walk draws the same state_0 (GNN.py:257) as the failed one
        import warnings
        warnings.warn('a bounded in-launch wait of the whole-loop kernels expired (GPU shared with other long-running work?): '
                      're-running these forwards with one launch per iteration', RuntimeWarning, stacklevel=3)
        saved = (self.native_flags, self.group_batches, self.inference_streams)
        self.native_flags = (self.native_flags & ~nat.FLAG_FUSED_GEN_MASK) | nat.FLAG_FUSED_GEN2
        self.group_batches, self.inference_streams = False, 1
        try:
            self._k_seen = []
            res = forwards()
            self._check_k()
        finally:
            self.native_flags, self.group_batches, self.inference_streams = saved
        self.recovered_walks = getattr(self, 'recovered_walks', 0) + 1
        return res

    def _check_k(self):
        """k < 0 is how the loop kernels report an expired in-launch wait (the persistent whole-loop kernel's grid barrier,
        the wave-specialised kernel's slot hand-off): checked once per predict() / evaluate(), where the host
        synchronises anyway."""
        ks = getattr(self, '_k_seen', None)
        if ks:
            if float(torch.stack([k.min() for k in ks]).min()) < 0:
                raise nat.NativeError(self._K_ERROR)
        self._k_seen = None

    def check_last_k(self):
        """For direct `Loop()` / `call()` / `model(x)` callers: synchronise on the k of the most recent forward and raise
        `NativeError` if the kernels reported an expired wait (k < 0). Returns k as a float."""
        k = getattr(self, '_last_k', None)
        if k is None: raise RuntimeError('no forward has run yet')
        kv = k.detach().cpu().numpy()
        if kv.min() < 0: raise nat.NativeError(self._K_ERROR)
        return float(kv) if kv.ndim == 0 else kv

    @staticmethod
    def _batch_device(x):
        t = x[0]
        return t.device if isinstance(t, torch.Tensor) else torch.device('cpu')

    # batch overlap and mask-index cache -------------------------------------------------------------------------------
    group_batches = True           # predict() / evaluate(): merged batches as independent loops of one launch where supported

    def _forward_batches(self, sequencer, device):
        """(i, out_i) for every batch of `sequencer`, in order.  Small batches leave most of the GPU idle (a merged MUTAG
        batch is 15 workgroups on 256 CUs), so where the library supports it consecutive batches are merged into one graph
        and run as INDEPENDENT loops of one launch (`Loop(groups=...)`: every batch keeps its own convergence test and
        iteration count, so each result equals the batch-by-batch call); otherwise batches are spread over side streams."""
        n = len(sequencer)
        plan = self._group_plan(sequencer, device) if n > 1 else None
        if plan is None:
            yield from self._batches_concurrently(n, lambda i: self.call(sequencer[i][0], training=False), device,
                                                  self._round_width(sequencer, device) if n else 1)
            return
        outs = {}
        for li, out in self._run_plan(plan, lambda li: self._plan_launch(sequencer, plan[li]), device):
            r0 = 0
            for i in plan[li]:
                rows = int(sequencer[i][1].shape[0])               # output rows of a batch = its target rows
                outs[i] = out[r0: r0 + rows]
                r0 += rows
        for i in range(n): yield i, outs[i]

    def _plan_launch(self, sequencer, batches):
        """The output rows of one plan entry: its batches merged and run as convergence groups of one call.  A merge the library
        refuses after all (the plan looks at shapes; e.g. an operand form only the call sees) is rerun batch by batch."""
        parts = getattr(batches, 'parts', None)
        if len(batches) == 1 and not parts: return self.call(sequencer[batches[0]][0], training=False)
        x, node_begin = sequencer.merged_batches(batches)
        sets = None
        if parts:
            node_begin, sets = batches.groups_and_sets({b: node_begin[i + 1] - node_begin[i] for i, b in enumerate(batches)})
        try:
            return self.call(x, training=False, groups=node_begin, **({'group_sets': sets} if sets else {}))
        except (RuntimeError, nat.NativeError) as e:
            if 'convergence groups' not in str(e): raise
            return torch.cat([self.call(sequencer[b][0], training=False) for b in batches], dim=0)

    def _run_plan(self, plan, launch, device):
        """(li, launch(li)) for every entry of a group plan, on side streams.  Entries whose loop kernel has a grid barrier (the
        spread form, single batches: every workgroup of the launch must be resident at once, and two such launches that
        become ready together could each hold CUs the other is waiting for) all go to ONE stream and run one after the other;
        the barrier-free resident launches (one workgroup per group, `_Launch.resident`) overlap them on the other streams.
        `inference_streams` <= 1 runs everything in order on the caller's stream."""
        width = max(1, min(len(plan), 4, int(self.inference_streams)))
        nxt = [0]
        def lane(li):
            if width <= 2 or not getattr(plan[li], 'resident', False): return 0
            nxt[0] += 1
            return 1 + (nxt[0] - 1) % (width - 1)
        if width == 2: lane = lambda li: 1 if getattr(plan[li], 'resident', False) else 0
        yield from self._batches_concurrently(len(plan), launch, device, width, lane=lane)

    def _group_plan(self, sequencer, device):
        """[[batch, ...], ...]: the batches each launch merges, or None when grouping does not apply (composite model, CPU,
        'normalized', unsupported shape, a sequencer that does not keep a list of merged batches).  Batches whose state fits the
        LDS of one CU go together - one workgroup each, any number of them per launch; the others in runs of at most 32 whose
        64-node tiles are all resident at once."""
        if not self.group_batches or device.type != 'cuda' or not hasattr(sequencer, 'merged_batches'): return None
        if not isinstance(getattr(sequencer, 'graph_tensors', None), list): return None          # opt-in: Multi* sequencers only
        if not isinstance(getattr(self, 'net_state', None), Sequential) or not hasattr(self, 'state_vect_dim'): return None
        if sequencer.merged_batches(0, 1) is None: return None
        # the plan depends on the batches (rebuilt batches = a new graph_tensors list) and on the model's shape only: kept
        key = (id(sequencer.graph_tensors), len(sequencer), self.state_vect_dim, self.max_iteration, self.native_flags, self._focus,
               tuple(self.net_state.units), tuple(self.net_state.activations), self.net_state.input_dim, str(device))
        cache = self.__dict__.setdefault('_plan_cache', {})
        hit = cache.get(key)
        if hit is not None and hit[0] is sequencer.graph_tensors: return hit[1]
        plan = self._make_group_plan(sequencer, device)
        if len(cache) >= 4: cache.clear()
        cache[key] = (sequencer.graph_tensors, plan)
        return plan

    def _make_group_plan(self, sequencer, device):
        try:
            sizes = [int(sequencer[i][0][0].shape[0]) for i in range(len(sequencer))]
            n_out = [int(sequencer[i][1].shape[0]) for i in range(len(sequencer))]
            x0 = sequencer[0][0]
            L, A = int(x0[0].shape[1]), int(x0[1].shape[1]) - 2
            # hub rows (in-degree > sparse.HEAVY_THRESHOLD) take the per-iteration kernels: such a batch is launched alone
            hub = [SparseMatrix.from_triple(sequencer[i][0][5]).device_csr(device).get('heavy') is not None for i in range(len(sequencer))]
        except Exception:
            return None
        focus = nat.FOCUS[self._focus]
        supported = lambda bs: ops.loop_groups_supported(int(sum(sizes[b] for b in bs)), L, A, self.net_state, self.net_output,
                                                         self.state_vect_dim, self.max_iteration, focus, self.native_flags,
                                                         sum(n_out[b] for b in bs), [0] + [int(v) for v in np.cumsum([sizes[b] for b in bs])])
        plan, rest = [_Launch([b]) for b in range(len(sizes)) if hub[b]], [b for b in range(len(sizes)) if not hub[b]]
        if not rest: return plan
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        biggest = max(rest, key=lambda b: sizes[b])
        if supported([min(rest, key=lambda b: sizes[b])]) == 2:      # one CU per batch, its state in LDS - for those that fit
            fits = set(rest if supported([biggest]) == 2 else [b for b in rest if supported([b]) == 2])
            # a batch that does not fit is cut along graph boundaries into parts that do; the parts run on one CU each and share only
            # the loop's condition (group sets): still ONE launch, and no batch is left to the spread form
            parts = {}
            S_w = self.state_vect_dim if self.state_vect_dim > 0 else L
            limit = (158 * 1024) // (4 * (16 if S_w <= 16 else 32) + 16)          # kernel_state_lds.hpp: LDS_BUDGET_BYTES / (row + CSR record)
            for b in rest:
                if b in fits: continue
                cut = self._cut_batch(sequencer, b, limit)
                if cut is not None: parts[b] = cut
            cand = [b for b in rest if b in fits or b in parts]
            # One workgroup = one CU per group: with fewer groups than CUs the launch leaves CUs idle while its largest group sets
            # the pace.  Cut the largest batches further (two, three .. balanced parts) until the groups fill the GPU - when the loop
            # is long enough to pay for the flag exchange the parts then make every iteration (MUTAG, 4 337 graphs: d = 32 x 50
            # iterations 1.32 -> 1.18 ms per walk, the starter configuration's 16-wide x 5 iterations 0.31 -> 0.42 ms: not cut).
            n_all = sum(len(parts.get(b, [0])) for b in cand)
            long_loop = self.max_iteration * max(sizes[b] for b in cand) * (16 if S_w <= 16 else 32) >= 500_000 if cand else False
            if len(cand) == len(rest) and n_all < cus and long_loop and os.environ.get('GNN_FILL_CUS', '1') != '0':
                cur = {b: parts.get(b, [sizes[b]]) for b in cand}
                while n_all < cus:
                    b = max(cand, key=lambda b_: max(cur[b_]))
                    cut = self._cut_batch(sequencer, b, limit, len(cur[b]) + 1)
                    if cut is None or max(cut) >= max(cur[b]): break
                    cur[b] = cut; n_all += 1
                parts = {b: v for b, v in cur.items() if len(v) > 1}
            chunk, n_grp = [], 0
            def flush():
                nonlocal chunk, n_grp, rest
                if len(chunk) >= 2 or (chunk and chunk[0] in parts):
                    entry = _Launch(chunk, resident=True, parts={b: parts[b] for b in chunk if b in parts})
                    begin, sets = entry.groups_and_sets(sizes)
                    if ops.loop_groups_supported(begin[-1], L, A, self.net_state, self.net_output, self.state_vect_dim, self.max_iteration, focus,
                                                 self.native_flags, sum(n_out[b] for b in chunk), begin, sets) == 2:
                        plan.append(entry)
                        rest = [b for b in rest if b not in set(chunk)]
                chunk, n_grp = [], 0
            any_cut = bool(parts)
            for b in cand:
                g_b = len(parts.get(b, [0]))
                if chunk and n_grp + g_b > (cus if any_cut else 1024): flush()
                chunk.append(b); n_grp += g_b
            flush()
        # A spread run needs every one of its 64-node tiles resident at once (grid barrier), one workgroup per CU; a resident launch
        # holds one CU per group.  Size the spread runs for the CUs the resident launch leaves free, so both kinds really run side
        # by side (a run sized for the whole GPU would spin at its first barrier until the resident launch has drained).
        n_res = sum(len(bs.groups_and_sets(sizes)[0]) - 1 for bs in plan if getattr(bs, 'resident', False))      # one CU per GROUP (a cut batch holds several)
        free = cus - min(n_res, cus)
        biggest_rest = max([(sizes[b] + 63) // 64 for b in rest], default=0)
        if free >= max(biggest_rest, cus // 4): cus = free
        run, tiles = [], 0
        for b in rest:
            t = (sizes[b] + 63) // 64
            if run and (tiles + t > cus or len(run) >= 32):
                plan.append(_Launch(run)); run, tiles = [], 0
            run.append(b); tiles += t
        if run: plan.append(_Launch(run))
        for bs in plan:
            if len(bs) >= 2 and not getattr(bs, 'parts', None) and not supported(bs): return None
        return plan

    @staticmethod
    def _cut_batch(sequencer, b, limit, n_parts=None):
        """Node counts of the parts batch `b` is cut into - contiguous runs of whole graphs, each of at most `limit` nodes, `n_parts`
        of them (default: as few as fit), balanced (the parts run side by side, the largest sets the pace) - or None when that is
        not possible (a graph larger than `limit`, fewer graphs than parts)."""
        sizes_g = [int(g.nodes.shape[0]) for g in sequencer.data[b * sequencer.batch_size: (b + 1) * sequencer.batch_size]]
        if not sizes_g or max(sizes_g) > limit: return None
        total = sum(sizes_g)
        for p_try in range(n_parts or max(1, -(-total // limit)), len(sizes_g) + 1):
            target = -(-total // p_try)
            parts, cur = [], 0
            for i, n in enumerate(sizes_g):
                left_parts = p_try - len(parts) - 1                      # parts still to open after the current one
                # close the current part when it has reached its share (and enough graphs remain to fill the others), or when
                # the next graph would not fit
                if cur and (cur + n > limit or (cur + n // 2 >= target and left_parts > 0 and len(sizes_g) - i >= left_parts)):
                    parts.append(cur); cur = 0
                cur += n
            parts.append(cur)
            if max(parts) <= limit and (n_parts is None or len(parts) == n_parts): return parts
            if n_parts is not None: return None
        return None

    def _batches_concurrently(self, n, fn, device, width=None, lane=None):
        """Run fn(i), i < n, `width` at a time on side HIP streams and yield (i, result) in order on the caller's stream.

        Batches of a sequencer are independent graphs; a merged MUTAG batch keeps ~16 of the 256 CUs busy (the whole loop
        is one persistent launch of one workgroup per 64 nodes), so several run side by side: batch i goes to
        stream i % width, the caller's stream waits for all of them before the first result is used, and every result is
        recorded on the caller's stream so the caching allocator does not hand its memory back early. `width` is capped so that the
        persistent kernels of one round can all be resident at once (they wait for each other inside the launch).  `lane`
        (optional): i -> stream index in [0, width) instead of the round-robin."""
        width = int(width or self.inference_streams)
        if device.type != 'cuda' or width <= 1 or n <= 1:
            for i in range(n): yield i, fn(i)
            return
        if self._streams is None or len(self._streams) < width or self._streams[0].device != device:
            self._streams = [torch.cuda.Stream(device) for _ in range(width)]
        main = torch.cuda.current_stream(device)
        for st in self._streams[:width]: st.wait_stream(main)          # inputs made on the caller's stream are ready
        res = []
        for i in range(n):                                             # round-robin: at most `width` loops in flight,
            with torch.cuda.stream(self._streams[(lane(i) if lane else i) % width]):   # each stream runs its batches in order
                res.append(fn(i))
        for st in self._streams[:width]: main.wait_stream(st)
        for i, out in enumerate(res):
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.is_cuda: t.record_stream(main)
            yield i, out

    def _round_width(self, sequencer, device):
        """How many batches of `sequencer` may run side by side: every persistent loop kernel of a round must be resident
        at once (one workgroup per 64 nodes, one workgroup per CU)."""
        if device.type != 'cuda': return 1
        try:
            n_max = max(int(sequencer[i][0][0].shape[0]) for i in range(len(sequencer)))
        except Exception:
            return 1
        tiles = max(1, (n_max + 63) // 64)
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        return max(1, min(int(self.inference_streams), cus // (tiles + 2)))

    def _out_index(self, set_mask, output_mask):
        if set_mask.is_cuda:
            from ..device_batch import lookup_out_index            # batches assembled on the device bring their index along
            pre = lookup_out_index(set_mask, output_mask)
            if pre is not None: return pre
        key = (set_mask.data_ptr(), set_mask._version, output_mask.data_ptr(), output_mask._version, len(set_mask))
        hit = self._mask_cache.get(key)
        if hit is None:
            idx = torch.nonzero(torch.logical_and(set_mask, output_mask)).reshape(-1).to(torch.int32)
            if len(self._mask_cache) > 4096: self._mask_cache.clear()
            hit = self._mask_cache[key] = (idx, set_mask, output_mask)   # keep the masks alive: ptr is the key
        return hit[0]


def _squeeze_last(x):
    return x.squeeze(-1) if isinstance(x, torch.Tensor) and x.dim() > 1 and x.shape[-1] == 1 else x


def _arc_endpoints(adjacency: SparseMatrix, device):
    key = ('endpoints', str(device))
    if key not in adjacency._dev:
        idx = torch.from_numpy(adjacency.indices.astype(np.int32)).to(device)
        adjacency._dev[key] = (idx[:, 0].contiguous(), idx[:, 1].contiguous())
    return adjacency._dev[key]


#######################################################################################################################
class GNNnodeBased(_LoopModel):
    """GNN for node-focused problems (reference GNN.py:8-306)."""
    name = "node"
    _focus = 'n'

    def __init__(self, net_state: Sequential, net_output: Sequential, state_vect_dim: int, max_iteration: int,
                 state_threshold: float) -> None:
        assert state_vect_dim >= 0
        assert max_iteration >= 0
        assert state_threshold >= 0
        self.net_state = net_state
        self.net_output = net_output
        self.state_vect_dim = int(state_vect_dim)
        self.max_iteration = int(max_iteration)
        self.state_threshold = state_threshold
        self.native_flags = 0          # OR of _native.FLAG_* (tests use FLAG_UNFUSED)
        self.loop_events = None        # optional (begin, end) torch.cuda.Event pair recorded around the iterations
        self._engine_init()

    # ---- copy / config / persistence ---------------------------------------------------------------------------------
    def copy(self, copy_weights: bool = True):
        config = self.get_config()
        config["net_state"] = config["net_state"].clone(copy_weights)
        config["net_output"] = config["net_output"].clone(copy_weights)
        return self.from_config(config)

    def get_config(self):
        return {"net_state": self.net_state, "net_output": self.net_output, "state_vect_dim": self.state_vect_dim,
                "max_iteration": self.max_iteration, "state_threshold": self.state_threshold}

    @classmethod
    def from_config(cls, config, **kwargs):
        return cls(**config)

    def __repr__(self):
        return f"GNN(type={self.name}, state_dim={self.state_vect_dim}, " \
               f"threshold={self.state_threshold}, max_iter={self.max_iteration}), avg={self.average_st_grads}"

    __str__ = __repr__

    @staticmethod
    def _save_net(net: Sequential, folder: str):
        os.makedirs(folder, exist_ok=True)
        np.savez(os.path.join(folder, 'weights.npz'), *net.get_weights())
        with open(os.path.join(folder, 'config.json'), 'w') as f: json.dump(net.get_config(), f)

    @staticmethod
    def _load_net(folder: str) -> Sequential:
        with open(os.path.join(folder, 'config.json')) as f: cfg = json.load(f)
        data = np.load(os.path.join(folder, 'weights.npz'))
        return Sequential(**cfg, weights=[data[f'arr_{i}'] for i in range(len(data.files))])

    def save(self, path: str, *args, **kwargs):
        """`<path>/net_state/`, `<path>/net_output/`, `<path>/config.json` with the reference's keys (GNN.py:94-115);
        each network is weights.npz (Keras get_weights() order) + its layer config."""
        if path[-1] != '/': path += '/'
        config = self.get_config()
        self._save_net(config.pop("net_state"), f'{path}net_state/')
        self._save_net(config.pop("net_output"), f'{path}net_output/')
        with open(f'{path}config.json', 'w') as json_file: json.dump(config, json_file)

    @classmethod
    def load(cls, path: str, *args, **kwargs):
        if path[-1] != '/': path += '/'
        with open(f'{path}config.json', 'r') as read_file: config = json.loads(read_file.read())
        return cls(net_state=cls._load_net(f'{path}net_state/'), net_output=cls._load_net(f'{path}net_output/'), **config)

    def summary(self, *args, **kwargs):
        print(repr(self))
        for net in [self.net_state, self.net_output]:
            print('\n')
            net.summary(*args, **kwargs)

    # ---- call ---------------------------------------------------------------------------------------------------------
    def call(self, inputs, training: bool = False, mask=None, *, groups=None, group_sets=None):
        """`inputs` = the list a sequencer's `__getitem__` yields; returns `out` (eval) or `(k, state, out)`.  `groups`
        (additive): node offsets of merged batches that run as independent loops (see `Loop`)."""
        inputs = self.process_inputs(inputs)
        kw = {} if groups is None else {'groups': groups}
        if group_sets is not None: kw['group_sets'] = group_sets
        k, state, out = self.Loop(*inputs, training=training, **kw)
        if training: return k, state, out
        if getattr(self, '_k_seen', None) is not None: self._k_seen.append(k)        # predict() / evaluate() check it at the end
        return out

    @staticmethod
    def process_inputs(inputs):
        """Squeeze [2] dim_node_label, [3] set_mask, [4] output_mask; turn [5:] triples into `SparseMatrix`
        (reference GNN.py:181-193 rebuilds tf.SparseTensor there; here the cached CSR is reused)."""
        inputs = list(inputs)
        inputs[2:5] = [_squeeze_last(k) for k in inputs[2:5]]
        inputs[5:] = [SparseMatrix.from_triple(k) for k in inputs[5:]]
        return inputs

    # ---- pieces of the loop, callable on their own (tests, LGNN-style callers) ----------------------------------------
    def condition(self, k, state, state_old, *args):
        """Device-side predicate of reference GNN.py:196-214 (`torch.ops.gnnkeras.converged`); returns a 0-dim bool tensor
        (no host sync)."""
        nat.require_device(state, 'state')
        flag = ops.converged(state.to(torch.float32).contiguous(), None if state_old is None else state_old.to(torch.float32).contiguous(),
                             self.state_threshold)
        kk = k if isinstance(k, torch.Tensor) else torch.tensor(float(k), device=state.device)
        return torch.logical_and(flag[0] != 0, kk.to(state.device) < self.max_iteration)

    def convergence(self, k, state, state_old, nodes, adjacency, aggregated_nodes, aggregated_arcs, training, *,
                    arcs=None, arcnode=None):
        """One state-transition step with the reference's own eight positional arguments (reference GNN.py:217-236):

            state_new = net_state([state | nodes (if state_vect_dim > 0) | adjacency^T . state | aggregated_nodes | aggregated_arcs])

        through `torch.ops.gnnkeras.state_step` -> `gnn_state_step_agg`: the aggregates `Loop` formed once (GNN.py:254-258) are
        used as handed in, exactly as the reference threads them through `tf.while_loop`; `training=True` runs the state network
        in training mode (BatchNormalization on the batch statistics of this call + moving-average update, Dropout) on the training
        primitives.  `arcs=` / `arcnode=` (additive, optional): when BOTH aggregates are None they are rebuilt from these on the
        device.  Returns the reference's 8-tuple."""
        nat.require_device(nodes, 'nodes'); nat.require_device(state, 'state')
        dev = nodes.device
        training = bool(training)
        nodes32 = nodes.to(torch.float32).contiguous()
        state32 = state.to(dev, torch.float32).contiguous()
        adj = SparseMatrix.from_triple(adjacency).device_csr(dev)
        if aggregated_nodes is None and aggregated_arcs is None:
            if arcs is None or arcnode is None:
                raise ValueError('convergence() needs aggregated_nodes / aggregated_arcs (the reference\'s arguments), or arcs= and arcnode= to rebuild them')
            arcs32 = arcs.to(dev, torch.float32).contiguous()
            aggregated_arcs = ops.aggregate(SparseMatrix.from_triple(arcnode).device_csr(dev), arcs32[:, 2:].contiguous())
            aggregated_nodes = ops.aggregate(adj, nodes32) if self.state_vect_dim > 0 else nodes32.new_zeros((nodes32.shape[0], 0))
        N = nodes32.shape[0]
        agg_n = nodes32.new_zeros((N, 0)) if aggregated_nodes is None else aggregated_nodes.to(dev, torch.float32)
        agg_a = nodes32.new_zeros((N, 0)) if aggregated_arcs is None else aggregated_arcs.to(dev, torch.float32)
        if agg_n.stride(-1) != 1: agg_n = agg_n.contiguous()
        if agg_a.stride(-1) != 1: agg_a = agg_a.contiguous()
        if training and (self.net_state.batch_normalization or self.net_state.dropout_rate):
            from .training import mlp_training_call
            segs = [(state32, None)]
            if self.state_vect_dim > 0: segs.append((nodes32, None))
            segs.append((ops.aggregate(adj, state32), None))
            if agg_n.shape[1] > 0: segs.append((agg_n, None))
            if agg_a.shape[1] > 0: segs.append((agg_a, None))
            new = mlp_training_call(self.net_state, segs, N)
        else:
            dummy_arcs = nodes32.new_zeros((0, 2 + agg_a.shape[1]))
            new, _moving = ops.state_step(nodes32, dummy_arcs, adj, None, self.net_state, state32, self.state_vect_dim,
                                          self.state_threshold, self.native_flags, aggregated=(agg_n, agg_a))
        return k + 1, new, state, nodes, adjacency, aggregated_nodes, aggregated_arcs, training

    def apply_filters(self, state_converged, nodes, adjacency, arcs_label, mask):
        """Rows of [state | labels] (or state) where mask (reference GNN.py:239-242). In `Loop` this gather is fused
        into the output network's first layer; this standalone version exists for API parity."""
        if self.state_vect_dim: state_converged = torch.cat([state_converged, nodes], dim=1)
        return state_converged[mask]

    # ---- the loop -----------------------------------------------------------------------------------------------------
    def Loop(self, nodes, arcs, dim_node_label, set_mask, output_mask, adjacency, arcnode, nodegraph,
             training: bool = False, *, state0=None, seed=None, node_level: bool = False, groups=None, group_sets=None):
        """(k, state, out) for one (merged) graph — reference GNN.py:245-274.

        `groups` (additive; inference only): node offsets [G + 1] of G batches merged into this graph.  The loop then runs
        as G independent loops in one launch - each batch has its own `condition` and stops on its own, exactly as if the
        reference had been called batch by batch - and k is a vector of G iteration counts (include/gnnloop.h,
        group_node_begin; `ops.loop_groups_supported` says whether a shape qualifies).  `group_sets` (first-group offsets [B + 1]):
        a batch too big for one CU's LDS is cut along graph boundaries into several groups that share only the loop's condition -
        they stop together, exactly like the uncut batch - and k has one entry per SET (= batch).

        Additive keyword arguments: `state0` replaces the reference's `tf.random.normal(stddev=0.1)`
        draw when `state_vect_dim > 0`; otherwise it is drawn on the device with `seed`. `node_level=True` makes a
        graph-focused model return its per-node outputs (what the reference obtains by calling the unbound
        `GNNnodeBased.Loop` on a graph-based model inside LGNN, LGNN.py:225)."""
        focus = 'n' if (node_level and self._focus == 'g') else self._focus
        if training:
            if groups is not None: raise ValueError('groups are an inference-only feature')
            from .training import LoopTrainer
            if getattr(self, '_trainer', None) is None: self._trainer = LoopTrainer(self)      # (one per model, shared with train_step)
            x_list = [nodes, arcs, dim_node_label, set_mask, output_mask, adjacency, arcnode, nodegraph]
            if self._trainer._native_forward_applies():
                # one library call (include/gnnloop.h ABI 9, forward_only) instead of ~ 40 building-block calls: what a serial LGNN fit()
                # runs on every single graph between its layers (reference LGNN.py:325-337)
                try:
                    k, state, out = self._trainer.forward_native(x_list, state0=state0, seed=seed, node_level=node_level)
                    return torch.tensor(float(k), device=state.device), state, out
                except nat.NativeError as e:
                    # an expired grid barrier of the persistent forward kernel (GPU shared with long-running work) or a shape the in-library
                    # forward does not cover: nothing has been touched - the building blocks below have no cross-workgroup waits
                    if not any(t in str(e) for t in ('grid barrier', 'cannot be resident', 'train through the building blocks', 'empty graph')): raise
            tp = self._trainer.forward(x_list, state0=state0, seed=seed, node_level=node_level)
            return torch.tensor(float(tp.k), device=tp.dev), tp.state.clone(), tp.y_pred
        nat.require_device(nodes, 'nodes'); nat.require_device(arcs, 'arcs')
        dev = nodes.device
        set_mask, output_mask = _squeeze_last(set_mask).to(dev), _squeeze_last(output_mask).to(dev)
        out_index = self._out_index(set_mask, output_mask)
        N = nodes.shape[0]
        if self.state_vect_dim > 0:
            if state0 is None:
                gen = None
                if seed is not None:
                    gen = torch.Generator(device=dev); gen.manual_seed(int(seed))
                state0 = torch.empty((N, self.state_vect_dim), device=dev, dtype=torch.float32).normal_(0.0, 0.1, generator=gen)      # tf.random.normal(stddev=0.1), one launch
            state0 = state0.to(dev, torch.float32).contiguous()
            if tuple(state0.shape) != (N, self.state_vect_dim): raise ValueError('state0 must be (n_nodes, state_vect_dim)')
        else:
            state0 = None
        adjacency = SparseMatrix.from_triple(adjacency)
        adj, arcn = adjacency.device_csr(dev), SparseMatrix.from_triple(arcnode).device_csr(dev)
        ends = _arc_endpoints(adjacency, dev) if focus == 'a' else None
        ng = SparseMatrix.from_triple(nodegraph).device_csr(dev) if focus == 'g' else None
        # (the first group of every set, uploaded BEFORE the launch: a pageable host-to-device copy behind it would block the host
        # until the loop has finished)
        set_id = None
        if group_sets is not None:
            gs_ = np.asarray([int(v) for v in group_sets], dtype=np.int64)
            set_id = torch.as_tensor(np.repeat(np.arange(len(gs_) - 1), np.diff(gs_)), device=dev)
        # the whole Loop is ONE custom op: torch.ops.gnnkeras.loop_forward (csrc/torch_ops.cpp -> gnn_loop_forward)
        k, state, out = ops.loop_forward(nodes.to(torch.float32).contiguous(), arcs.to(torch.float32).contiguous(), adj, arcn, ng,
                                         self.net_state, self.net_output, state0, out_index, ends, self.state_vect_dim,
                                         self.max_iteration, self.state_threshold, nat.FOCUS[focus], self.native_flags,
                                         loop_events=self.loop_events, groups=groups, group_sets=group_sets)
        if group_sets is not None:
            # one entry per set: its groups report the same k - or -1e9 where a member's wait for the others expired (any member: the
            # minimum keeps it, so _check_k / check_last_k see it)
            k = torch.full((len(group_sets) - 1,), float('inf'), device=dev).scatter_reduce_(0, set_id, k, 'amin')
        self._last_k = k
        return k, state, out


#######################################################################################################################
class GNNarcBased(GNNnodeBased):
    """GNN for arc-focused problems (reference GNN.py:312-330): output network sees
    [state_src | label_src | state_dst | label_dst | arc label] of every masked arc."""
    name = "arc"
    _focus = 'a'

    def apply_filters(self, state_converged, nodes, adjacency, arcs_label, mask):
        if self.state_vect_dim: state_converged = torch.cat([state_converged, nodes], dim=1)
        adjacency = SparseMatrix.from_triple(adjacency)
        idx = torch.from_numpy(adjacency.indices).to(state_converged.device)
        states = state_converged[idx].reshape(arcs_label.shape[0], 2 * state_converged.shape[1])
        return torch.cat([states, arcs_label], dim=1)[mask]


#######################################################################################################################
class GNNgraphBased(GNNnodeBased):
    """GNN for graph-focused problems (reference GNN.py:336-346): per-graph mean of the node outputs through
    NodeGraph, pooled on the device as one more CSR walk."""
    name = "graph"
    _focus = 'g'
