"""Small-graph data sets over the GPUs of one node: batch-parallel predict() / evaluate(), data-parallel fit().

SURVEY.md §8e "Other cases": a merged MUTAG batch is block-diagonal (reference graph_class.py:399-408), so it shards by GRAPH
with no halo at all.  The reference is single-process; what must be reproduced is its result:

  * `predict` / `evaluate` (Keras semantics over the sequencer, reference GNN.py:165-177): the launches of the model's group
    plan are dealt round-robin to the ranks, outputs are all-gathered (predict) or the loss / metric sums all-reduced (evaluate);
  * `fit` / `train_step` (reference GNN.py:277-306): every batch of the sequencer is cut into one shard of whole graphs per rank
    and the step reproduces the SINGLE-PROCESS step on the whole batch - not an average of R independent steps:
      - BatchNormalization in training mode uses the statistics of ALL nodes of the merged batch (MLP.py:67-70):
        per BN layer and iteration the ranks exchange (count, mean, variance) of their rows and combine them exactly;
      - the loop's `condition` is a `reduce_any` over all nodes (GNN.py:212): the per-iteration flag is all-reduced, so every
        rank runs the same k iterations;
      - the loss is SUM_OVER_BATCH_SIZE over all target rows; weight gradients are sums over all rows: the first layer's
        P = X^T dZ, q = colsum(dZ) are all-reduced per network call (its BatchNorm input gradient needs the global moments),
        the deeper layers' gradients once per step.
    All ranks then apply the same update to the same weights: no parameter broadcast after initialisation.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  Collectives are issued on
device tensors and are stream-ordered; the step has the same single host synchronisation as the single-process one (reading k)
plus one for the row counts.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .Models.GNN import History, _loss_fn, _metric_fn


class DPContext:
    """The collectives of one data-parallel training step (used by `Models.training.LoopTrainer` when given as `dp=`)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._counts = {}

    # ---- row counts ------------------------------------------------------------------------------------------------------
    def begin_step(self, n_nodes: int, n_out: int, type_counts=None):
        """Every rank learns every shard's node / output-row counts (one small all-gather, one host read).  `type_counts`
        (heterogeneous models): the shard's nodes per type - a type may have no node at all on some ranks, its network then takes part
        in the collectives with weight zero (Models/training.py)."""
        dev = self._device
        mine = torch.tensor([n_nodes, n_out] + [int(c) for c in (type_counts or [])], dtype=torch.int64, device=dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        c = torch.stack(parts).cpu().numpy()
        self._counts = {'nodes': c[:, 0].tolist(), 'out': c[:, 1].tolist()}
        for t in range(len(type_counts or [])): self._counts[f'type{t}'] = c[:, 2 + t].tolist()
        if min(self._counts['nodes']) == 0 or min(self._counts['out']) == 0:
            raise ValueError('data-parallel step: a rank holds no nodes / no output rows (fewer graphs than ranks, or a shard with '
                             f'every node masked out): {self._counts}')

    def set_device(self, device):
        self._device = torch.device(device)

    def total_rows(self, kind: str, local: int = None) -> int:
        if kind not in self._counts:
            t = torch.tensor([int(local)], dtype=torch.int64, device=self._device)
            dist.all_reduce(t, group=self.group)
            self._counts[kind] = [int(t)]
        return int(sum(self._counts[kind]))

    # ---- collectives -----------------------------------------------------------------------------------------------------
    def combine_stats(self, mean: torch.Tensor, var: torch.Tensor, kind: str):
        """(mean, biased variance) of the rows of ALL shards from the per-shard ones: with n = sum n_r,
        mean = sum n_r mean_r / n, var = sum n_r (var_r + (mean_r - mean)^2) / n - exact, no E[x^2] - E[x]^2 cancellation;
        evaluated in float64 in rank order on every rank (identical bits everywhere)."""
        K = mean.shape[0]
        mine = torch.cat([mean.reshape(-1), var.reshape(-1)])
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        allp = torch.stack(parts).to(torch.float64)                          # [R, 2K]
        n = torch.tensor(self._counts[kind], dtype=torch.float64, device=mine.device)[:, None]
        tot = n.sum().clamp(min=1.0)
        mu = (n * allp[:, :K]).sum(0) / tot
        va = (n * (allp[:, K:] + (allp[:, :K] - mu) ** 2)).sum(0) / tot
        return mu.to(torch.float32), va.to(torch.float32)

    def all_reduce_sum(self, *tensors):
        """Sum over the ranks, in place; several tensors travel as ONE flat buffer."""
        if len(tensors) == 1 and tensors[0].is_contiguous():
            dist.all_reduce(tensors[0], group=self.group)
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.all_reduce(flat, group=self.group)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view(t.shape)); off += t.numel()

    def any_flag(self, flag: torch.Tensor):
        """flag = max over the ranks (int32 words; the `reduce_any` of the reference's condition across shards)."""
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous, balanced share [lo, hi) of n_items for `rank` (sizes differ by at most one)."""
    return n_items * rank // world, n_items * (rank + 1) // world


class DataParallel:
    """`DataParallel(model).fit / evaluate / predict(sequencer)`: the Keras-style calls of the wrapped model, collectively over
    the ranks of `group`.  Every rank constructs the same model (same weights) and the same sequencer (same graphs, same order)
    and calls the same method; results are identical on every rank.

    `exact=True` (default): a training step reproduces the single-process step on the whole batch (statistics, loop condition and
    gradients span the ranks; the collectives sit between the iteration's launches, so it runs the building-block orchestration).
    `exact=False` ("replicas", SURVEY 8e's trivial fallback and what data-parallel training usually means): every rank runs the
    WHOLE in-library step (`gnn_train_step`: the persistent kernels on MUTAG-sized shards) on its shard alone - its own batch
    statistics, its own iteration count - and one all-reduce averages the gradients weighted by target rows (plus the
    BatchNormalization moving statistics): not the reference's step on the merged batch, an order of magnitude less device time."""

    def __init__(self, model, group=None, exact: bool = True):
        # (heterogeneous models - reference CompositeGNN.py:275-304 - train in both modes since round 5: the exact step all-gathers the
        # batch statistics of every type's network with the shard's row count of that type as weight, zero included)
        self.exact = bool(exact)
        if self.exact:
            # (refused HERE, on every rank alike - not from inside a step, where a rank without rows of the network would
            #  already sit in the step's all-reduce while the others raise)
            nets = list(model.net_state) if isinstance(model.net_state, (list, tuple)) else [model.net_state]
            for n_ in nets + [model.net_output]:
                if any(float(r) > 0 and int(q) == 0 for r, q in zip(n_.dropout_rate or [], n_.dropout_pos or [])):
                    raise NotImplementedError('the exact data-parallel step does not take a Dropout layer in front of the first Dense (dropout_pos = 0): '
                                              'use DataParallel(model, exact=False)')
        self.model, self.group = model, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.dp = DPContext(group)
        self._trainer = None

    # ---- shards of a batch ---------------------------------------------------------------------------------------------------
    def shard(self, sequencer, index: int):
        """(x_list, targets, sample_weight) of this rank's graphs of batch `index`: a contiguous share of whole graphs, merged like
        any batch."""
        if not hasattr(sequencer, 'shard_item'):
            raise TypeError(f'{type(sequencer).__name__} cannot be sharded by graph (a list of graphs per batch is required)')
        return sequencer.shard_item(index, self.rank, self.world)

    # ---- training ------------------------------------------------------------------------------------------------------------
    def train_step(self, data, *, state0=None, seed=None, apply=True):
        """One optimisation step on ONE batch of which `data` is this rank's shard (collective).  Returns {'loss', 'k', metrics}:
        the numbers of the whole batch, the same on every rank."""
        from .Models.training import LoopTrainer
        m = self.model
        if m.loss is None: raise RuntimeError('compile() the model with a loss before fit() / train_step()')
        if self._trainer is None: self._trainer = LoopTrainer(m, dp=self.dp if self.exact else None)
        x, y, sample_weight = data
        self.dp.set_device(x[0].device)
        self.dp._counts = {}
        if not self.exact:
            res = self._replica_step(x, y, sample_weight, state0, seed, apply)
        else:
            res = self._trainer.train_step(x, y, sample_weight, state0=state0, seed=seed, apply=apply)
        out = {'loss': res['loss'], 'k': res['k'], 'y_pred': res['y_pred']}
        if y is not None and m.metrics_spec:
            yd = y.to(res['y_pred'].device)
            sw = torch.ones(yd.shape[0], device=yd.device) if sample_weight is None else sample_weight.to(yd.device)
            sums = torch.stack([(f(yd, res['y_pred']) * sw).sum() for _, f in (_metric_fn(mm, yd.shape[-1]) for mm in m.metrics_spec)] + [sw.sum()])
            self.dp.all_reduce_sum(sums)
            for i, mm in enumerate(m.metrics_spec): out[_metric_fn(mm, yd.shape[-1])[0]] = sums[i] / sums[-1]
        return out

    def _replica_step(self, x, y, sample_weight, state0, seed, apply):
        """The single-process step on this rank's shard, then ONE all-reduce: gradients and loss weighted by the shard's share of the
        batch's target rows (each rank's loss is a mean over its own rows), BatchNormalization moving statistics averaged, k = the
        largest of the ranks' iteration counts."""
        tr, m = self._trainer, self.model
        res = tr.train_step(x, y, sample_weight, state0=state0, seed=seed, apply=False)
        for attempt in range(2):
            gs = tr.gs if isinstance(tr.gs, (list, tuple)) else [tr.gs]
            pairs = [gv for g_ in list(gs) + [tr.go] for gv in zip(g_.gradients(), g_.variables())]
            grads = [g for g, _ in pairs]
            moving = [t for g_ in list(gs) + [tr.go] if g_.bn for t in g_.moving]
            dev = grads[0].device
            rows = float(y.shape[0]) if y is not None else 0.0
            head = torch.tensor([rows, 0.0, float(res['k'])], dtype=torch.float32, device=dev)
            head[1] = res['loss'].detach().reshape(()) * rows
            # the in-library step's validity word (1: gradients valid; 0: this rank's persistent backward launch lost a grid barrier, its
            # gradients are NaN and its moving statistics untouched) travels with the sums: a rank that failed repeats ITS shard on the
            # building blocks and the all-reduce is done again - nobody's update ever sees the poisoned values
            okv = res['grads_ok'].to(torch.float32).reshape(1) if res.get('grads_ok') is not None else torch.ones(1, dtype=torch.float32, device=dev)
            flat = torch.cat([head[:2]] + [torch.where(okv > 0, g.reshape(-1) * rows, torch.zeros_like(g.reshape(-1))) for g in grads] +
                             [t.reshape(-1) for t in moving] + [okv])
            kmax = head[2:3].clone()
            if self.world > 1:
                dist.all_reduce(flat, group=self.group)
                dist.all_reduce(kmax, op=dist.ReduceOp.MAX, group=self.group)
            if int(round(float(flat[-1].item()))) >= self.world or attempt == 1: break
            if float(okv.item()) == 0.0:
                import warnings
                warnings.warn('the persistent backward kernel of this rank\'s training step could not keep its workgroups resident: the shard is '
                              'trained again on the general kernels', RuntimeWarning, stacklevel=3)
                res = tr._train_step_general(x, y, sample_weight, state0, seed, False)
        flat = flat[:-1]
        total = flat[0].clamp(min=1.0)
        off = 2
        for g in grads:
            n = g.numel(); g.copy_((flat[off:off + n] / total).view_as(g)); off += n
        for t in moving:
            n = t.numel(); t.copy_((flat[off:off + n] / self.world).view_as(t)); off += n
        res = dict(res, loss=flat[1] / total, k=int(kmax.item()))
        self._rows_total = float(flat[0])
        if apply: m._optimizer_obj().apply_gradients(pairs)
        return res

    def fit(self, sequencer, epochs: int = 1, validation_data=None, verbose: int = 1, callbacks=None, initial_epoch: int = 0, **kwargs):
        """The single-process `fit` with every batch sharded over the ranks (same batches, same order, same updates, the same
        callback protocol: `set_model`, `on_train_begin`, `on_epoch_begin`, `on_epoch_end`, `on_train_end`, `initial_epoch`)."""
        m = self.model
        m._check_kwargs('fit', kwargs)
        cbs = list(callbacks or [])
        for cb in cbs:
            if hasattr(cb, 'set_model'): cb.set_model(m)
        def emit(name, *args):
            for cb in cbs:
                f = getattr(cb, name, None)
                if f is not None: f(*args)
        history = History()
        m.stop_training = False
        emit('on_train_begin', {})
        logs = {}
        for epoch in range(int(initial_epoch), epochs):
            emit('on_epoch_begin', epoch, {})
            tot, wsum = {}, 0.0
            for i in range(len(sequencer)):
                data = self.shard(sequencer, i)
                r = self.train_step(data)
                w = float(self.dp.total_rows('targets')) if self.exact else self._rows_total
                for key, val in r.items():
                    if key in ('k', 'y_pred'): continue
                    tot[key] = tot.get(key, 0.0) + float(val) * w
                wsum += w
            logs = {key: val / max(wsum, 1.0) for key, val in tot.items()}
            if validation_data is not None:
                logs.update({'val_' + key: val for key, val in self.evaluate(validation_data, return_dict=True).items()})
            for key, val in logs.items(): history.setdefault(key, []).append(val)
            history.epoch.append(epoch)
            if verbose and self.rank == 0:
                print(f'Epoch {epoch + 1}/{epochs} - ' + ' - '.join(f'{k_}: {v:.4f}' for k_, v in logs.items()))
            emit('on_epoch_end', epoch, logs)
            if hasattr(sequencer, 'on_epoch_end'): self._synchronised_epoch_end(sequencer)
            if m.stop_training: break
        emit('on_train_end', logs)
        m.history = history
        return history

    def _synchronised_epoch_end(self, sequencer):
        """Every rank must reshuffle its copy of the data set the same way: rank 0 draws a seed, everyone shuffles with it (the
        caller's own numpy stream is left where it was)."""
        seed = torch.tensor([np.random.randint(0, 2 ** 31 - 1)], dtype=torch.int64, device=self.dp._device if hasattr(self.dp, '_device') else 'cpu')
        dist.broadcast(seed, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        state = np.random.get_state()
        np.random.seed(int(seed))
        try:
            sequencer.on_epoch_end()
        finally:
            np.random.set_state(state)

    # ---- inference -----------------------------------------------------------------------------------------------------------
    def _my_outputs(self, sequencer, device):
        """{batch: output rows} of the launches this rank runs: entry li of the model's group plan goes to rank li % world (without
        a plan: batch i to rank i % world)."""
        m = self.model
        n = len(sequencer)
        plan = m._group_plan(sequencer, device) if n > 1 else None
        owner = self._owner_of(n, plan)
        my_batches = [i for i in range(n) if owner[i] == self.rank]
        my_launches = None if plan is None else [plan[li] for li in range(len(plan)) if li % self.world == self.rank]

        def run():
            outs = {}
            if my_launches is None or not m.group_batches:
                # (no plan - or the recovery walk of `_with_recovery`, which switches grouping off: the SAME batches, one by one)
                for j, out in m._batches_concurrently(len(my_batches), lambda j: m.call(sequencer[my_batches[j]][0], training=False), device,
                                                      m._round_width(sequencer, device) if my_batches else 1):
                    outs[my_batches[j]] = out
            else:
                for li, out in m._run_plan(my_launches, lambda li: m._plan_launch(sequencer, my_launches[li]), device):
                    r0 = 0
                    for i in my_launches[li]:
                        rows = int(sequencer[i][1].shape[0])
                        outs[i] = out[r0:r0 + rows]; r0 += rows
            return outs
        # an expired cross-workgroup wait on THIS rank's launches is repaired on this rank (no collective in here)
        return m._with_recovery(run, device), plan

    def _owner_of(self, n, plan):
        if plan is None: return [i % self.world for i in range(n)]
        owner = [0] * n
        for li, bs in enumerate(plan):
            for b in bs: owner[b] = li % self.world
        return owner

    def predict(self, sequencer, **kwargs):
        """Outputs of every batch, concatenated in batch order, on every rank (Keras `predict` semantics)."""
        m = self.model
        m._check_kwargs('predict', kwargs)
        n = len(sequencer)
        if n == 0: return np.zeros((0, 0), np.float32)
        device = m._batch_device(sequencer[0][0])
        self.dp.set_device(device)
        outs, plan = self._my_outputs(sequencer, device)
        rows = [int(sequencer[i][1].shape[0]) for i in range(n)]
        owner = self._owner_of(n, plan)
        T = m.net_output.units[-1]
        per_rank = [sum(rows[i] for i in range(n) if owner[i] == r) for r in range(self.world)]
        pad = max(max(per_rank), 1)
        mine = torch.zeros((pad, T), dtype=torch.float32, device=device)
        r0 = 0
        for i in range(n):
            if owner[i] == self.rank:
                mine[r0:r0 + rows[i]] = outs[i]; r0 += rows[i]
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        parts = [p.cpu().numpy() for p in parts]
        pos = [0] * self.world
        result = []
        for i in range(n):
            r = owner[i]
            result.append(parts[r][pos[r]:pos[r] + rows[i]]); pos[r] += rows[i]
        return np.concatenate(result, axis=0)

    def evaluate(self, sequencer, return_dict: bool = False, **kwargs):
        """Loss and metrics over the sequencer (Keras `evaluate` semantics, as `_LoopModel.evaluate`): every rank evaluates the
        batches it ran, the weighted sums are all-reduced."""
        m = self.model
        m._check_kwargs('evaluate', kwargs)
        if m.loss is None: raise RuntimeError('compile() the model with a loss before evaluate()')
        n = len(sequencer)
        if n == 0: raise ValueError('evaluate() needs at least one batch')
        device = m._batch_device(sequencer[0][0])
        self.dp.set_device(device)
        outs, plan = self._my_outputs(sequencer, device)
        lossf = _loss_fn(m.loss)
        T_y = int(sequencer[0][1].shape[-1])
        mets = [_metric_fn(mm, T_y) for mm in m.metrics_spec]
        sums = torch.zeros(3 + len(mets), dtype=torch.float64, device=device)       # loss sum, rows, weight sum, metric sums
        for i, p in sorted(outs.items()):
            y, sw = sequencer[i][1].to(device), sequencer[i][2].to(device)
            sums[0] += (lossf(y, p) * sw).sum(); sums[1] += sw.shape[0]; sums[2] += sw.sum()
            for j, (_, f) in enumerate(mets): sums[3 + j] += (f(y, p) * sw).sum()
        dist.all_reduce(sums, group=self.group)
        res = {'loss': float(sums[0] / sums[1])}
        for j, (name, _) in enumerate(mets): res[name] = float(sums[3 + j] / sums[2])
        return res if return_dict else [res['loss']] + [res[name] for name, _ in mets]
