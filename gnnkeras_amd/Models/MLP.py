"""State / output networks of the GNN: the weights container the HIP kernels consume.

Mirror of the reference's `GNN/Models/MLP.py`: `MLP(...)` builds the same stack the reference builds with Keras —
an optional `BatchNormalization` first (default ON, `MLP.py:14, 67-70`), then `Dense` x n, Dropout / AlphaDropout at
the requested positions — and `get_inout_dims(...)` computes the same input / layer widths (`MLP.py:82-140`).
There is no Keras here: `Sequential` below is a plain container whose weights live in HBM as float32 torch tensors in
Keras `get_weights()` order (BN: gamma, beta, moving_mean, moving_variance; Dense: kernel[in,out], bias[out]) and
whose forward pass is `gnn_mlp_forward` of libgnnloop.so (MFMA f32).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Union

import numpy as np
import torch

from .. import _native as nat
from ..sparse import default_device

BN_EPSILON = 1e-3       # tf.keras.layers.BatchNormalization default
BN_MOMENTUM = 0.99
_TRUNC = 0.87962566103423978   # stddev of a unit normal truncated to [-2, 2]


# ----------------------------------------------------------------------------------------------------------------------
# Keras initializers by name (the reference passes the strings straight to `Dense`, starter.py:23-30)
# ----------------------------------------------------------------------------------------------------------------------
def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    return shape[0], shape[1]


def _truncated_normal(rng, shape, stddev):
    out = rng.normal(0.0, 1.0, size=shape)
    bad = np.abs(out) > 2.0
    while np.any(bad):
        out[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * stddev


def initialize(name, shape, rng) -> np.ndarray:
    """numpy restatement of the Keras initializer `name` (VarianceScaling family uses a truncated normal whose
    stddev is corrected by 1/0.8796...)."""
    if callable(name):
        return np.asarray(name(shape), dtype=np.float32)
    name = 'zeros' if name is None else str(name).lower()
    fan_in, fan_out = _fans(shape)
    if name == 'zeros': return np.zeros(shape, dtype=np.float32)
    if name == 'ones': return np.ones(shape, dtype=np.float32)
    if name == 'lecun_normal': return _truncated_normal(rng, shape, np.sqrt(1.0 / fan_in) / _TRUNC).astype(np.float32)
    if name == 'glorot_normal': return _truncated_normal(rng, shape, np.sqrt(2.0 / (fan_in + fan_out)) / _TRUNC).astype(np.float32)
    if name == 'he_normal': return _truncated_normal(rng, shape, np.sqrt(2.0 / fan_in) / _TRUNC).astype(np.float32)
    if name == 'glorot_uniform':
        lim = np.sqrt(6.0 / (fan_in + fan_out)); return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    if name == 'lecun_uniform':
        lim = np.sqrt(3.0 / fan_in); return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    if name == 'he_uniform':
        lim = np.sqrt(6.0 / fan_in); return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    if name == 'random_normal': return rng.normal(0.0, 0.05, size=shape).astype(np.float32)
    if name == 'random_uniform': return rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
    raise ValueError(f'unknown initializer {name!r}')


# ----------------------------------------------------------------------------------------------------------------------
class L1L2:
    """Weight penalty `l1 * sum|w| + l2 * sum(w^2)` (the semantics of `tf.keras.regularizers.L1L2`, which the reference
    passes through `MLP(..., kernel_regularizer=, bias_regularizer=)`, MLP.py:12-15, :48-49): added once per variable to
    the training loss (`train_step`: `regularization_losses=self.losses`, GNN.py:286) and to its gradient."""

    def __init__(self, l1: float = 0.0, l2: float = 0.0):
        self.l1, self.l2 = float(l1), float(l2)

    def get_config(self):
        return {'l1': self.l1, 'l2': self.l2}

    def __repr__(self):
        return f'L1L2(l1={self.l1}, l2={self.l2})'


def l1(l: float = 0.01): return L1L2(l1=l)
def l2(l: float = 0.01): return L1L2(l2=l)
def l1_l2(l1: float = 0.01, l2: float = 0.01): return L1L2(l1=l1, l2=l2)


def as_regularizer(r):
    """None | L1L2 | 'l1' / 'l2' / 'l1_l2' (Keras string shortcuts, factor 0.01) | {'l1':, 'l2':} | any object with float
    `l1` / `l2` attributes (a Keras L1L2 instance); anything else is rejected rather than ignored."""
    if r is None or isinstance(r, L1L2): return r
    if isinstance(r, str):
        name = r.lower()
        if name in ('l1', 'l2', 'l1_l2'): return {'l1': l1, 'l2': l2, 'l1_l2': l1_l2}[name]()
        raise ValueError(f'unknown regularizer {r!r}')
    if isinstance(r, dict): return L1L2(r.get('l1', 0.0) or 0.0, r.get('l2', 0.0) or 0.0)
    if hasattr(r, 'l1') or hasattr(r, 'l2'): return L1L2(float(getattr(r, 'l1', 0.0) or 0.0), float(getattr(r, 'l2', 0.0) or 0.0))
    raise ValueError(f'unsupported regularizer {r!r}: only L1 / L2 weight penalties have a device gradient')


# ----------------------------------------------------------------------------------------------------------------------
class Sequential:
    """[BatchNormalization] + Dense x n with weights resident on the device.

    Surface kept from `tf.keras.models.Sequential` as far as the reference uses it: `get_weights`, `set_weights`,
    `trainable_variables`, `name`, `summary`, call with `training=`; `input_dim`, `units`, `activations` describe it."""

    def __init__(self, input_dim: int, units, activations, batch_normalization: bool = True, dropout_rate=(),
                 dropout_pos=(), alphadropout: bool = False, name: Optional[str] = None, weights=None, device=None,
                 kernel_regularizer=None, bias_regularizer=None):
        self.input_dim = int(input_dim)
        self.units = [int(u) for u in units]
        self.activations = [('linear' if a is None else str(a).lower()) for a in activations]
        for a in self.activations:
            if a not in nat.ACTIVATIONS: raise ValueError(f'unsupported activation {a!r}')
        if not (1 <= len(self.units) <= nat.GNN_MAX_LAYERS):
            raise ValueError(f'between 1 and {nat.GNN_MAX_LAYERS} Dense layers are supported')
        self.batch_normalization = bool(batch_normalization)
        self.dropout_rate, self.dropout_pos, self.alphadropout = list(dropout_rate), list(dropout_pos), alphadropout
        n = len(self.units)
        kr = kernel_regularizer if isinstance(kernel_regularizer, (list, tuple)) else [kernel_regularizer] * n
        br = bias_regularizer if isinstance(bias_regularizer, (list, tuple)) else [bias_regularizer] * n
        if len(kr) != n or len(br) != n: raise ValueError('one regularizer per Dense layer (or one for all)')
        self.kernel_regularizer, self.bias_regularizer = [as_regularizer(r) for r in kr], [as_regularizer(r) for r in br]
        self.name = name
        self.device = torch.device(device) if device is not None else default_device()
        self._weights: list[torch.Tensor] = []
        self._native = None
        if weights is not None:
            self.set_weights(weights)

    # ---- weights -----------------------------------------------------------------------------------------------------
    def weight_shapes(self):
        shapes = []
        if self.batch_normalization: shapes += [(self.input_dim,)] * 4
        fan_in = self.input_dim
        for u in self.units:
            shapes += [(fan_in, u), (u,)]
            fan_in = u
        return shapes

    def set_weights(self, weights):
        shapes = self.weight_shapes()
        if len(weights) != len(shapes):
            raise ValueError(f'expected {len(shapes)} weight arrays, got {len(weights)}')
        new = []
        for w, s in zip(weights, shapes):
            t = w.detach().to(torch.float32) if isinstance(w, torch.Tensor) else torch.as_tensor(np.asarray(w, dtype=np.float32))
            if tuple(t.shape) != tuple(s): raise ValueError(f'weight shape {tuple(t.shape)} != {s}')
            new.append(t.to(self.device).contiguous().clone())
        self._weights = new
        self._native = None

    def get_weights(self):
        return [w.detach().cpu().numpy().copy() for w in self._weights]

    @property
    def weights(self): return list(self._weights)

    @property
    def trainable_variables(self):
        """Dense kernels/biases and BN gamma/beta (moving statistics are not trainable), Keras order."""
        if not self.batch_normalization: return list(self._weights)
        return self._weights[0:2] + self._weights[4:]

    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self.device = device
            self._weights = [w.to(device) for w in self._weights]
            self._native = None
        return self

    def spec(self):
        """(spec, weights) in the form the oracle's `mlp_apply` takes (tests only)."""
        reg = lambda rs: [None if r is None else (r.l1, r.l2) for r in rs]
        return {'batch_normalization': self.batch_normalization, 'activations': list(self.activations),
                'kernel_regularizer': reg(self.kernel_regularizer), 'bias_regularizer': reg(self.bias_regularizer),
                'dropout_rate': list(self.dropout_rate), 'dropout_pos': list(self.dropout_pos),
                'alphadropout': bool(self.alphadropout)}, self.get_weights()

    def get_config(self):
        return {'input_dim': self.input_dim, 'units': self.units, 'activations': self.activations,
                'batch_normalization': self.batch_normalization, 'dropout_rate': self.dropout_rate,
                'dropout_pos': self.dropout_pos, 'alphadropout': self.alphadropout, 'name': self.name,
                'kernel_regularizer': [None if r is None else r.get_config() for r in self.kernel_regularizer],
                'bias_regularizer': [None if r is None else r.get_config() for r in self.bias_regularizer]}

    def clone(self, copy_weights: bool = True, rng=None, kernel_initializer='glorot_uniform', bias_initializer='zeros'):
        m = Sequential(**self.get_config(), device=self.device)
        if copy_weights: m.set_weights(self._weights)
        else: m.set_weights(_init_weights(m, kernel_initializer, bias_initializer, rng or np.random.default_rng()))
        return m

    def summary(self, *args, **kwargs):
        print(f'Sequential "{self.name}": input {self.input_dim}')
        if self.batch_normalization: print(f'  BatchNormalization({self.input_dim})')
        for u, a in zip(self.units, self.activations): print(f'  Dense({u}, activation={a})')
        print(f'  parameters: {sum(int(np.prod(s)) for s in self.weight_shapes())}')

    # ---- native view -------------------------------------------------------------------------------------------------
    def native(self) -> nat.MLP:
        """ctypes `gnn_mlp_t` pointing at the device weights (kept alive by self)."""
        if self._native is None:
            if not self._weights: raise ValueError('network has no weights')
            m = nat.MLP()
            m.in_dim, m.n_layers = self.input_dim, len(self.units)
            pos = 0
            if self.batch_normalization:
                m.has_bn, m.bn_eps = 1, BN_EPSILON
                m.bn_gamma, m.bn_beta, m.bn_mean, m.bn_var = (nat.ptr(w) for w in self._weights[0:4])
                pos = 4
            for i, (u, a) in enumerate(zip(self.units, self.activations)):
                m.units[i], m.activation[i] = u, nat.ACTIVATIONS[a]
                m.kernel[i], m.bias[i] = self._weights[pos].data_ptr(), self._weights[pos + 1].data_ptr()
                pos += 2
            self._native = m
        return self._native

    def __call__(self, x: torch.Tensor, training: bool = False, *, seed=None):
        """Forward on the GPU.  Inference: `torch.ops.gnnkeras.mlp_forward` (moving statistics folded into the first Dense).
        `training=True` (Keras `net(x, training=True)`): BatchNormalization on the batch statistics of `x` with its moving averages
        moved once, Dropout / AlphaDropout with a fresh mask (`seed`, additive, makes it reproducible) - the training primitives
        of `Models/training.py`; a network without either layer computes the same thing in both modes."""
        nat.require_device(x, 'x')
        x = x.to(torch.float32)
        if training and (self.batch_normalization or self.dropout_rate):
            from .training import mlp_training_call
            if x.dim() != 2 or x.shape[1] != self.input_dim: raise ValueError(f'x must be [M, {self.input_dim}]')
            if x.stride(-1) != 1: x = x.contiguous()
            return mlp_training_call(self, [(x, None)], x.shape[0], seed=seed)
        from .. import ops
        return ops.mlp_forward(self, x.contiguous())


def _init_weights(model: Sequential, kernel_initializer, bias_initializer, rng):
    n = len(model.units)
    ki = kernel_initializer if isinstance(kernel_initializer, list) else [kernel_initializer] * n
    bi = bias_initializer if isinstance(bias_initializer, list) else [bias_initializer] * n
    w = []
    if model.batch_normalization:
        d = model.input_dim
        w += [np.ones(d, np.float32), np.zeros(d, np.float32), np.zeros(d, np.float32), np.ones(d, np.float32)]
    fan_in = model.input_dim
    for u, k, b in zip(model.units, ki, bi):
        w += [initialize(k, (fan_in, u), rng), initialize(b, (u,), rng)]
        fan_in = u
    return w


def MLP(input_dim: tuple, layers: list, activations, kernel_initializer, bias_initializer,
        kernel_regularizer=None, bias_regularizer=None, dropout_rate: Union[list, float, None] = None,
        dropout_pos: Optional[Union[list, int]] = None, alphadropout: bool = False, batch_normalization: bool = True,
        *, name: str = None, rng=None, device=None) -> Sequential:
    """Same arguments as the reference builder (`MLP.py:12-15`); `rng` (numpy Generator or seed) is additive and makes
    the initial weights reproducible. Regularizers (`L1L2` / `l1()` / `l2()` of this module, Keras-style objects with
    `l1` / `l2` attributes, or the Keras string shortcuts) add their penalty to the training loss and gradients."""
    layers = [layers] if isinstance(layers, int) else list(layers)
    if type(activations) != list: activations = [activations for _ in layers]
    if type(kernel_initializer) != list: kernel_initializer = [kernel_initializer for _ in layers]
    if type(bias_initializer) != list: bias_initializer = [bias_initializer for _ in layers]
    if type(kernel_regularizer) != list: kernel_regularizer = [kernel_regularizer for _ in layers]
    if type(bias_regularizer) != list: bias_regularizer = [bias_regularizer for _ in layers]
    if type(dropout_pos) == int: dropout_pos = [dropout_pos]
    if type(dropout_rate) == float: dropout_rate = [dropout_rate for _ in dropout_pos]
    if dropout_rate is None or dropout_pos is None: dropout_rate, dropout_pos = list(), list()

    if len(set(map(len, [activations, kernel_initializer, bias_initializer, kernel_regularizer, bias_regularizer, layers]))) > 1:
        raise ValueError('Dense parameters must have the same length to be correctly processed')
    if len(dropout_rate) != len(dropout_pos):
        raise ValueError('Dropout parameters must have the same length to be correctly processed')

    in_dim = int(input_dim[0]) if isinstance(input_dim, (tuple, list)) else int(input_dim)
    model = Sequential(in_dim, layers, activations, batch_normalization, dropout_rate, dropout_pos, alphadropout,
                       name=None if name is None else name.lower(), device=device,
                       kernel_regularizer=kernel_regularizer, bias_regularizer=bias_regularizer)
    rng = rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)
    model.set_weights(_init_weights(model, kernel_initializer, bias_initializer, rng))
    return model


def get_inout_dims(net_name: str, dim_node_label, dim_arc_label: int, dim_target: int, focus: str, dim_state: int,
                   hidden_units: Optional[Union[int, list]] = None, *, layer: int = 0, get_state: bool = False,
                   get_output: bool = False):
    """Input shapes and layer widths of the state / output MLPs (reference `MLP.py:82-140`), including the LGNN
    layer > 0 rules. Returns ([ (in_dim,), ... one per node type ], [hidden..., out])."""
    assert layer >= 0
    assert focus in ['a', 'n', 'g']
    assert dim_state >= 0
    assert isinstance(hidden_units, (int, type(None))) or (isinstance(hidden_units, list) and all(isinstance(x, int) for x in hidden_units))

    NL, AL, T = np.array(dim_node_label, ndmin=1), dim_arc_label, dim_target
    DS, GS, GO = dim_state, get_state, get_output

    if layer > 0:
        if DS != 0:
            NL = NL + DS * GS + T * (focus != 'a') * GO
            AL = AL + T * (focus == 'a') * GO
        else:
            NL = NL + layer * NL * GS + ((layer - 1) * GS + 1) * T * (focus != 'a') * GO
            AL = AL + T * (focus == 'a') * GO

    if net_name == 'state':
        input_shape = list(NL + int(np.sum(NL)) + AL + 2 * DS)
        output_shape = DS if DS else NL
    elif net_name == 'output':
        if len(NL) > 1: NL = np.array([0])
        input_shape = list((focus == 'a') * (NL + AL + DS) + NL + DS)
        output_shape = T
    else:
        raise ValueError(':param net_name: not in [\'state\', \'output\']')

    input_shape = [(int(i),) for i in input_shape]
    if not hidden_units: hidden_units = list()
    if isinstance(hidden_units, int): hidden_units = [hidden_units]
    out = [int(i) for i in np.array(output_shape, ndmin=1)] if not np.isscalar(output_shape) else [int(output_shape)]
    return input_shape, hidden_units + out
