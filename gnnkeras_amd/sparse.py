"""Sparse operand of the message-passing loop: the stand-in for `tf.SparseTensor` on the MI355X path.

The reference moves three sparse matrices through its sequencer as `(indices[nnz,2] i64, values[nnz,1] f32,
dense_shape[2] i64)` triples (reference `GNN/Sequencers/GraphSequencers.py:108-110`) and rebuilds a
`tf.SparseTensor` from each in `process_inputs` (`GNN/Models/GNN.py:181-193`). Every use on the hot path is
`sparse_dense_matmul(A, X, adjoint_a=True)` = AᵀX, i.e. a *gather by destination*. `SparseMatrix` keeps the same COO
triple for API compatibility and caches, built once per batch, what the HIP kernels actually walk: the CSR of Aᵀ
(row pointer by destination, int32 source ids, optional per-arc weights) resident in HBM.
"""
from __future__ import annotations

import numpy as np
import torch


def canonical_device(device) -> torch.device:
    """torch.device with an explicit index ('cuda' -> 'cuda:<current>'): cache keys must not depend on how it was spelled."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        return torch.device('cuda', torch.cuda.current_device())
    return device


def default_device() -> torch.device:
    """`cuda:<current>` on a GPU box, CPU otherwise (host-logic tests only: the kernels never run on CPU)."""
    if torch.cuda.is_available():
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


class CSRByDestination:
    """CSR of Aᵀ for a COO matrix A (rows = sources / arcs, columns = destinations).

    rowptr : int32 [n_dst + 1]
    src    : int32 [nnz]   A-row index of every entry, grouped by destination; inside a destination the entries keep
                           the row-major (`tf.sparse.reorder`) order of A, i.e. ascending source — the accumulation
                           order of TF's CPU kernel for `adjoint_a=True`.
    w      : float32 [nnz] or None. None means "every entry of destination j has the same value row_scale[j]"
             (true for 'sum', 'average' and 'normalized': reference `graph_class.py:105-121`), which saves 4 B per arc
             per iteration of HBM traffic.
    row_scale : float32 [n_dst] or None (None with w None = all ones).
    """
    __slots__ = ('rowptr', 'src', 'w', 'row_scale', 'n_src', 'n_dst', 'nnz', 'max_degree')

    def __init__(self, rowptr, src, w, row_scale, n_src, n_dst):
        self.rowptr, self.src, self.w, self.row_scale = rowptr, src, w, row_scale
        self.n_src, self.n_dst, self.nnz = int(n_src), int(n_dst), int(len(src))
        self.max_degree = int(np.max(np.diff(rowptr))) if n_dst > 0 and len(rowptr) > 1 else 0

    @classmethod
    def from_coo(cls, rows, cols, values, shape, uniform_rows: bool = True):
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        cols = np.asarray(cols, dtype=np.int64).reshape(-1)
        values = np.asarray(values, dtype=np.float32).reshape(-1)
        n_src, n_dst = int(shape[0]), int(shape[1])
        if len(rows) and (rows.min() < 0 or rows.max() >= n_src or cols.min() < 0 or cols.max() >= n_dst):
            raise ValueError('sparse indices out of range of dense_shape')
        if n_src >= 2 ** 31 or len(rows) >= 2 ** 31:
            raise ValueError('graph too large for int32 ids')
        order = np.argsort(cols, kind='stable')
        counts = np.bincount(cols, minlength=n_dst)
        rowptr = np.zeros(n_dst + 1, dtype=np.int64)
        np.cumsum(counts, out=rowptr[1:])
        src = rows[order].astype(np.int32)
        w = values[order]
        row_scale = None
        if uniform_rows and len(w):
            first = np.ones(n_dst, dtype=np.float32)
            nz = counts > 0
            first[nz] = w[rowptr[:-1][nz]]
            if np.array_equal(w, np.repeat(first, counts)):
                row_scale = None if np.all(first == 1) else first
                w = None
        elif uniform_rows:
            w = None
        return cls(rowptr.astype(np.int32), src, w, row_scale, n_src, n_dst)


HEAVY_THRESHOLD = 512      # in-degree above which a destination row is aggregated by whole workgroups (hub nodes)
HEAVY_SEGMENT = 2048       # arcs per workgroup in that pre-pass


def split_heavy(c: CSRByDestination, threshold: int = HEAVY_THRESHOLD, segment: int = HEAVY_SEGMENT):
    """Skew handling for the fused iteration kernel, which gives 4..16 lanes to a destination row: a hub with 10^5
    in-arcs would serialise its whole tile. Rows with more than `threshold` arcs are cut into segments of <= `segment`
    arcs; a pre-pass kernel sums every segment with a whole workgroup into a *virtual source row* (index n_src + s), and
    the "light" operator the fused kernel walks lists those virtual rows in place of the hub's arcs.

    Returns (light CSRByDestination, heavy dict(rowptr int32[n_seg+1] into the ORIGINAL src / w arrays, n_seg)) or
    (c, None) when no row is heavy."""
    deg = np.diff(c.rowptr.astype(np.int64))
    heavy_rows = np.flatnonzero(deg > threshold)
    if len(heavy_rows) == 0:
        return c, None
    nseg_row = -(-deg[heavy_rows] // segment)
    n_seg = int(nseg_row.sum())
    # [seg_beg, seg_end) ranges inside the ORIGINAL src / w arrays, hub rows in ascending order, segments in arc order
    row_of_seg = np.repeat(heavy_rows, nseg_row)
    k_in_row = np.arange(n_seg) - np.repeat(np.cumsum(nseg_row) - nseg_row, nseg_row)
    seg_beg = c.rowptr[row_of_seg].astype(np.int64) + segment * k_in_row
    seg_end = np.minimum(seg_beg + segment, c.rowptr[row_of_seg + 1].astype(np.int64))
    # light operator: hub rows list their virtual sources
    new_deg = deg.copy()
    new_deg[heavy_rows] = nseg_row
    rowptr = np.zeros(c.n_dst + 1, dtype=np.int64)
    np.cumsum(new_deg, out=rowptr[1:])
    src = np.empty(int(rowptr[-1]), dtype=np.int32)
    w = None if c.w is None else np.ones(int(rowptr[-1]), dtype=np.float32)
    is_heavy = np.zeros(c.n_dst, dtype=bool); is_heavy[heavy_rows] = True
    keep = ~np.repeat(is_heavy, deg)                                # arcs of light rows, in order
    light_pos = np.repeat(~is_heavy, new_deg)
    src[light_pos] = c.src[keep]
    if w is not None: w[light_pos] = c.w[keep]
    virt = c.n_src + np.arange(n_seg, dtype=np.int64)
    src[~light_pos] = virt.astype(np.int32)                         # hub rows are ascending, so are their segments
    light = CSRByDestination(rowptr.astype(np.int32), src, w, c.row_scale, c.n_src + n_seg, c.n_dst)
    return light, dict(seg_beg=seg_beg.astype(np.int32), seg_end=seg_end.astype(np.int32), n_seg=n_seg)


class SparseMatrix:
    """COO triple with `tf.SparseTensor`'s attribute names (`indices`, `values`, `dense_shape`/`shape`), entries in
    canonical row-major order, plus the cached by-destination CSR (host and device copies).

    A matrix can also be *device-only* (`SparseMatrix.device_only`): assembled on the GPU (batch assembly,
    `gnnkeras_amd/device_batch.py`; `synth.er_device_batch`), it carries just the device CSR the kernels walk; the host COO
    (`indices` / `values`) is rebuilt from it on first access - tests and the oracle read it, the hot path never does."""

    def __init__(self, indices, values, dense_shape, *, reorder: bool = True):
        indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
        values = np.asarray(values, dtype=np.float32).reshape(-1)
        if len(indices) != len(values):
            raise ValueError('indices / values length mismatch')
        if reorder and len(values):
            order = np.lexsort((indices[:, 1], indices[:, 0]))
            if not np.array_equal(order, np.arange(len(order))):
                indices, values = indices[order], values[order]
        self._indices = indices
        self._values = values
        self.dense_shape = tuple(int(i) for i in np.asarray(dense_shape).reshape(-1))
        self._csr = None
        self._dev = {}

    @classmethod
    def device_only(cls, dense_shape, csr: dict, device, **more):
        """`csr`: dict(rowptr, src, w | None, row_scale | None, n_src, n_dst, nnz[, max_degree]) of tensors on `device`;
        `more`: further device-side forms to pre-seed the cache with (`by_source=dict`, `endpoints=(src, dst)`)."""
        m = object.__new__(cls)
        m._indices = m._values = None
        m.dense_shape = tuple(int(i) for i in dense_shape)
        m._csr = None
        d = dict(csr)
        d.setdefault('max_degree', 0); d.setdefault('light', None); d.setdefault('heavy', None)
        m._dev = {(str(canonical_device(device)), True): d}
        for key, val in more.items():
            if val is not None: m._dev[(key, str(canonical_device(device)))] = val
        return m

    def _materialise_host(self):
        """Host COO (row-major) of a device-only matrix, from its device CSR."""
        key = next(k for k in self._dev if isinstance(k[1], bool))
        d = self._dev[key]
        rowptr, src = d['rowptr'].cpu().numpy().astype(np.int64), d['src'].cpu().numpy().astype(np.int64)
        counts = np.diff(rowptr)
        dst = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
        if d['w'] is not None: val = d['w'].cpu().numpy().astype(np.float32)
        elif d['row_scale'] is not None: val = d['row_scale'].cpu().numpy().astype(np.float32)[dst]
        else: val = np.ones(len(src), dtype=np.float32)
        order = np.lexsort((dst, src))
        self._indices, self._values = np.stack([src[order], dst[order]], axis=1), val[order]

    # tf.SparseTensor-like surface -----------------------------------------------------------------------------------
    @property
    def indices(self):
        if self._indices is None: self._materialise_host()
        return self._indices

    @property
    def values(self):
        if self._values is None: self._materialise_host()
        return self._values

    @property
    def shape(self):
        return self.dense_shape

    @property
    def nnz(self):
        if self._values is None:
            return int(next(v for k, v in self._dev.items() if isinstance(k[1], bool))['nnz'])
        return len(self._values)

    @classmethod
    def from_scipy(cls, coo):
        """Reference `GraphTensor.COO2SparseTensor` (`graph_class.py:551-560`): empty matrix -> indices (0, 2)."""
        coo = coo.tocoo()
        if coo.size > 0:
            idx = np.stack([coo.row, coo.col], axis=1)
        else:
            idx = np.zeros((0, 2), dtype=np.int64)
        return cls(idx, coo.data, coo.shape)

    @classmethod
    def from_triple(cls, triple):
        """Accept what the sequencer emits: `(indices, values[nnz,1], dense_shape)`; torch or numpy."""
        if isinstance(triple, SparseMatrix):
            return triple
        if isinstance(triple, SparseTriple):
            return triple.matrix
        to_np = lambda x: x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        idx, val, shp = (to_np(t) for t in triple)
        return cls(idx, val.reshape(-1), shp.reshape(-1))

    def to_scipy(self):
        from scipy.sparse import coo_matrix
        return coo_matrix((self.values, (self.indices[:, 0], self.indices[:, 1])), shape=self.dense_shape,
                          dtype=np.float32)

    def copy(self):
        return SparseMatrix(self.indices.copy(), self.values.copy(), self.dense_shape, reorder=False)

    # diagonal blocks ------------------------------------------------------------------------------------------------
    def block_starts(self):
        """Rows at which a square matrix can be cut into diagonal blocks (no entry joins rows on different sides of a cut),
        as int64 [n_blocks + 1] from 0 to n: the graphs of a merged batch (reference graph_class.py:386-413) and whatever
        else happens to be disconnected AND contiguous.  From the host COO; a device-only matrix knows its blocks only when
        its assembler recorded them (`device_batch.py`), else None.  Cached."""
        b = getattr(self, '_blocks', None)
        if b is None:
            if self._indices is None or self.dense_shape[0] != self.dense_shape[1]: return None
            n = self.dense_shape[0]
            lo, hi = self._indices.min(axis=1), self._indices.max(axis=1)
            # cut c (between rows c - 1 and c) is crossed by an entry when lo < c <= hi
            crossed = np.cumsum(np.bincount(lo + 1, minlength=n + 2)[:n + 1] - np.bincount(hi + 1, minlength=n + 2)[:n + 1])
            b = self._blocks = np.append(np.flatnonzero(crossed[:n] == 0), n).astype(np.int64)
        return b

    def tiles(self, limit: int = 64):
        """Consecutive diagonal blocks packed greedily into tiles of at most `limit` rows: int32 [n_tiles + 1], or None when the
        blocks are unknown or one of them is larger than `limit`.  Cached per limit."""
        cache = self.__dict__.setdefault('_tiles', {})
        if limit not in cache:
            b = self.block_starts()
            t = None
            if b is not None and len(b) > 1 and int(np.diff(b).max()) <= limit:
                cuts, start = [0], 0
                ends = b[1:].tolist()
                for i, e in enumerate(ends):
                    if e - start > limit: start = ends[i - 1]; cuts.append(start)
                cuts.append(ends[-1])
                t = np.asarray(cuts, dtype=np.int32)
            cache[limit] = t
        return cache[limit]

    # by-destination CSR ---------------------------------------------------------------------------------------------
    def csr(self, uniform_rows: bool = True) -> CSRByDestination:
        if self._csr is None or self._csr[0] != uniform_rows:
            self._csr = (uniform_rows, CSRByDestination.from_coo(self.indices[:, 0], self.indices[:, 1], self.values,
                                                                 self.dense_shape, uniform_rows))
        return self._csr[1]

    def device_csr(self, device=None, uniform_rows: bool = True):
        """dict(rowptr, src, w|None, row_scale|None) of torch tensors on `device`, uploaded once and cached."""
        device = canonical_device(device) if device is not None else default_device()
        key = (str(device), uniform_rows)
        if key not in self._dev:
            if self._indices is None:
                raise ValueError(f'this matrix lives on {[k[0] for k in self._dev if isinstance(k[1], bool)]} only; asked for {device}')
            c = self.csr(uniform_rows)
            up = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)
            d = dict(rowptr=up(c.rowptr), src=up(c.src), w=up(c.w), row_scale=up(c.row_scale),
                     n_src=c.n_src, n_dst=c.n_dst, nnz=c.nnz, max_degree=c.max_degree, light=None, heavy=None)
            if c.max_degree > HEAVY_THRESHOLD:
                light, heavy = split_heavy(c)
                d['light'] = dict(rowptr=up(light.rowptr), src=up(light.src), w=up(light.w), row_scale=d['row_scale'],
                                  n_src=light.n_src, n_dst=light.n_dst, nnz=light.nnz)
                d['heavy'] = dict(seg_beg=up(heavy['seg_beg']), seg_end=up(heavy['seg_end']), n_seg=heavy['n_seg'])
            self._dev[key] = d
        return self._dev[key]

    def triple(self, device=None):
        """The `(indices, values[...,None], dense_shape)` tuple of `GraphSequencers.py:110`, carrying this matrix."""
        device = canonical_device(device) if device is not None else default_device()
        key = ('triple', str(device))
        if key not in self._dev:
            self._dev[key] = SparseTriple(self, device)
        return self._dev[key]

    def __repr__(self):
        return f'SparseMatrix(shape={self.dense_shape}, nnz={self.nnz})'


class SparseTriple(tuple):
    """3-tuple `(indices i64[nnz,2], values f32[nnz,1], dense_shape i64[2])` exactly as the reference sequencer emits
    it, that also remembers the `SparseMatrix` it came from so the model does not rebuild the CSR per call. The two big
    tensors are made on first access: the hot path (`SparseMatrix.from_triple`) never touches them."""

    def __new__(cls, matrix: SparseMatrix, device=None):
        device = torch.device(device) if device is not None else default_device()
        self = super().__new__(cls, (None, None, None))
        self.matrix, self._device, self._items = matrix, device, None
        return self

    def _make(self):
        if self._items is None:
            m = self.matrix
            self._items = (torch.from_numpy(m.indices).to(self._device), torch.from_numpy(m.values).to(self._device)[..., None],
                           torch.tensor(m.dense_shape, dtype=torch.int64))
        return self._items

    def __getitem__(self, i):
        return self._make()[i]

    def __iter__(self):
        return iter(self._make())

    def __len__(self):
        return 3

    def __repr__(self):
        return f'SparseTriple({self.matrix!r})'
