"""Host-side objects over the C-ABI: Context, DataView, State.

PyTorch is used here only as the owner of device memory (tensors hold the
columns, assignment vectors and score matrices) and of the HIP stream; every
computation is a call into libmicroscopes_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_NP_OF_TYPE = {L.TYPE_B: np.bool_, L.TYPE_I8: np.int8, L.TYPE_U8: np.uint8, L.TYPE_I16: np.int16,
               L.TYPE_U16: np.uint16, L.TYPE_I32: np.int32, L.TYPE_U32: np.uint32,
               L.TYPE_I64: np.int64, L.TYPE_U64: np.uint64, L.TYPE_F32: np.float32,
               L.TYPE_F64: np.float64}
_TORCH_OF_TYPE = {L.TYPE_B: torch.bool, L.TYPE_I8: torch.int8, L.TYPE_U8: torch.uint8,
                  L.TYPE_I16: torch.int16, L.TYPE_I32: torch.int32, L.TYPE_I64: torch.int64,
                  L.TYPE_F32: torch.float32, L.TYPE_F64: torch.float64}
_TYPE_OF_TORCH = {torch.bool: L.TYPE_B, torch.int8: L.TYPE_I8, torch.uint8: L.TYPE_U8,
                  torch.int16: L.TYPE_I16, torch.int32: L.TYPE_I32, torch.int64: L.TYPE_I64,
                  torch.float32: L.TYPE_F32, torch.float64: L.TYPE_F64}
for _n, _t in (("uint16", L.TYPE_U16), ("uint32", L.TYPE_U32), ("uint64", L.TYPE_U64)):
    if hasattr(torch, _n):
        _TYPE_OF_TORCH[getattr(torch, _n)] = _t
        _TORCH_OF_TYPE[_t] = getattr(torch, _n)

VALUE_TYPE = {L.BB: L.TYPE_B, L.BBNC: L.TYPE_B, L.GP: L.TYPE_U32, L.DD: L.TYPE_I32, L.NICH: L.TYPE_F32,
              L.NIW: L.TYPE_F32, L.NOOP: L.TYPE_B, L.BNB: L.TYPE_U32, L.DM: L.TYPE_I32}


def type_of_numpy(dt):
    """numpy scalar dtype -> primitive type (microscopes/common/_dataview.pyx:6-36)."""
    dt = np.dtype(dt)
    for t, npt in _NP_OF_TYPE.items():
        if np.dtype(npt) == dt:
            return t
    raise ValueError("Unknown type: %s" % dt)


def runtime_types_of(dtype):
    """structured dtype -> [(primitive type, count)] (_dataview.pyx:27-44)."""
    dtype = np.dtype(dtype)
    if len(dtype) == 0:
        raise ValueError("structural arrays only")
    out = []
    for i in range(len(dtype)):
        ft = dtype[i]
        if ft.subdtype is None:
            out.append((type_of_numpy(ft), 1))
        else:
            sub, shape = ft.subdtype
            if len(shape) != 1:
                raise ValueError("unsupported shape: %s" % (shape,))
            out.append((type_of_numpy(sub), int(shape[0])))
    return out


class Context(object):
    """One per process and GPU.  Kernels are enqueued on torch's current stream."""

    def __init__(self, device=0, stream=None):
        if not torch.cuda.is_available():
            raise L.MicroscopesHipError(-3, "no GPU visible to torch; common_amd has no CPU path")
        self.lib = L.load()
        self.device = int(device)
        torch.cuda.set_device(self.device)
        torch.cuda.init()
        self.torch_device = torch.device("cuda", self.device)
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        self._stream = s
        h = C.c_void_p()
        L.check(self.lib.msc_context_create(self.device, C.c_void_p(s.cuda_stream), C.byref(h)))
        self._h = h
        import weakref
        self._buffers = weakref.WeakSet()      # live msc_device_alloc* buffers that tensors alias (alloc / alloc_probed)

    def set_stream(self, stream):
        self._stream = stream
        L.check(self.lib.msc_context_set_stream(self._h, C.c_void_p(stream.cuda_stream)))

    def synchronize(self):
        L.check(self.lib.msc_context_synchronize(self._h))

    def build_info(self):
        return self.lib.msc_build_info().decode()

    def last_kernel(self, which="score"):
        """the kernel instantiation of this process's most recent scoring ("score") or fused assignment ("sweep") pass, as
        rocprofv3 spells it (msc_last_kernel): what bench.py keys the committed counter summaries by"""
        return self.lib.msc_last_kernel(0 if which == "score" else 1).decode()

    def value_op(self, family, dim, op, hp, ss_record, value=None):
        """One group::{add_value, remove_value, score_value, score_data} call (base.hpp:25-28) as a batch
        of one on the device.  op: "add" | "remove" | "score_value" | "score_data".  `ss_record` is a
        1-element array of ss_dtype(family, dim), updated in place by add/remove; returns the score."""
        code = {"add": 0, "remove": 1, "score_value": 2, "score_data": 3}[op]
        hpb = pack_hp(family, hp, dim)
        assert ss_record.dtype == ss_dtype(family, dim) and ss_record.size == 1 and ss_record.flags.c_contiguous
        vb = None
        if value is not None:
            vb = np.ascontiguousarray(value, dtype={L.TYPE_B: np.uint8, L.TYPE_U32: np.uint32, L.TYPE_I32: np.int32,
                                                    L.TYPE_F32: np.float32}[VALUE_TYPE[family]])
        score = C.c_float(0)
        L.check(self.lib.msc_value_op_single(self._h, int(family), int(dim), code, hpb.ctypes.data_as(C.c_void_p),
                                             ss_record.ctypes.data_as(C.c_void_p),
                                             vb.ctypes.data_as(C.c_void_p) if vb is not None else None,
                                             C.byref(score)))
        return float(score.value)

    def alloc(self, shape, dtype=torch.float32):
        """A zero-filled device tensor from msc_device_alloc -- the library's default allocator: from 64 MiB on the buffer
        is placed for the write stream of a score matrix (include/microscopes_hip.h).  The tensor owns the buffer."""
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        p = C.c_void_p()
        L.check(self.lib.msc_device_alloc(self._h, nbytes, C.byref(p)))
        return _alias_tensor(p.value, n, dtype, self.torch_device, owner=_DeviceBuffer(self, p.value)).reshape(*shape)

    def alloc_stats(self):
        """([GB/s fill rate of every candidate the most recent placed allocation tried], index kept)"""
        rates, n, chosen = (C.c_float * 64)(), C.c_uint32(), C.c_uint32()
        L.check(self.lib.msc_device_alloc_stats(self._h, rates, 64, C.byref(n), C.byref(chosen)))
        return [float(rates[i]) for i in range(min(n.value, 64))], int(chosen.value)

    def alloc_probed(self, shape, dtype=torch.float32, candidates=8):
        """A zero-filled device tensor in the best-placed of `candidates` allocations (msc_device_alloc_probed; the
        write stream of a large score matrix runs 5.6 or 7.0 TB/s depending on where the driver put it).
        -> (tensor, [GB/s of every candidate], index kept).  The tensor owns the buffer (freed with it)."""
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        p, rates, chosen = C.c_void_p(), (C.c_float * candidates)(), C.c_uint32()
        L.check(self.lib.msc_device_alloc_probed(self._h, nbytes, candidates, C.byref(p), rates, C.byref(chosen)))
        t = _alias_tensor(p.value, n, dtype, self.torch_device, owner=_DeviceBuffer(self, p.value)).reshape(*shape)
        return t, [float(r) for r in rates], int(chosen.value)

    def close(self):
        """destroy the context.  Refused while tensors from alloc() / alloc_probed() are alive: destroying the context
        unmaps their memory (the buffers hold a reference to the context, so garbage collection never gets here first)"""
        if getattr(self, "_h", None):
            live = [b for b in getattr(self, "_buffers", ()) if b.ptr]
            if live:
                raise RuntimeError("Context.close(): %d device buffer(s) from alloc()/alloc_probed() are still referenced "
                                   "by tensors; drop those first" % len(live))
            self.lib.msc_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DataView(object):
    """Columnar device copy of a packed recarray (replaces recarray numpy_dataview)."""

    def __init__(self, ctx, handle, keepalive=None):
        self.ctx, self._h, self._keep = ctx, handle, keepalive
        n, f = C.c_uint64(), C.c_uint32()
        L.check(ctx.lib.msc_dataview_size(handle, C.byref(n), C.byref(f)))
        self.nrows, self.nfeatures = n.value, f.value

    @classmethod
    def from_recarray(cls, ctx, npd, col_types=None):
        """numpy structured (optionally masked) 1-D array, as recarray/_dataview.pyx:61-92."""
        if npd is None:
            raise ValueError("npd is None")
        if len(npd.shape) != 1:
            raise ValueError("1D (structural) arrays only")
        types = runtime_types_of(npd.dtype)
        if hasattr(npd, "mask"):
            data = np.ascontiguousarray(npd.data)
            mask = np.ascontiguousarray(np.ma.getmaskarray(npd))
            mask = mask.view(np.uint8).reshape(-1)
        else:
            data, mask = np.ascontiguousarray(npd), None
        if data.dtype.itemsize != sum(np.dtype(_NP_OF_TYPE[t]).itemsize * c for t, c in types):
            data = np.ascontiguousarray(data.astype(np.dtype([("f%d" % i, _NP_OF_TYPE[t], (c,)) if c > 1
                                                              else ("f%d" % i, _NP_OF_TYPE[t])
                                                              for i, (t, c) in enumerate(types)])))
        rt = (L.RuntimeType * len(types))(*[L.RuntimeType(t, c) for t, c in types])
        ct = None
        if col_types is not None:
            ct = (C.c_int32 * len(types))(*[int(t) for t in col_types])
        h = C.c_void_p()
        L.check(ctx.lib.msc_dataview_from_records(
            ctx._h, data.ctypes.data_as(C.c_void_p),
            mask.ctypes.data_as(C.c_void_p) if mask is not None else None,
            data.shape[0], rt, len(types), ct, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_tensors(cls, ctx, columns, masks=None):
        """Adopt device tensors as columns (no copy).  [N] scalars or [N, d] vector features."""
        cols = []
        for c in columns:
            if c.device != ctx.torch_device or not c.is_contiguous():
                raise ValueError("columns must be contiguous tensors on %s" % ctx.torch_device)
            cols.append(c)
        n = cols[0].shape[0]
        types = []
        for c in cols:
            if c.shape[0] != n or c.dim() > 2:
                raise ValueError("column shapes must be [N] or [N, d] with one N")
            types.append((_TYPE_OF_TORCH[c.dtype], 1 if c.dim() == 1 else int(c.shape[1])))
        rt = (L.RuntimeType * len(types))(*[L.RuntimeType(t, k) for t, k in types])
        ptrs = (C.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
        mptr = None
        if masks is not None:
            mptr = (C.c_void_p * len(cols))(*[(m.data_ptr() if m is not None else None) for m in masks])
        h = C.c_void_p()
        L.check(ctx.lib.msc_dataview_from_device_columns(ctx._h, n, rt, len(types), ptrs, mptr, C.byref(h)))
        return cls(ctx, h, keepalive=(cols, masks))

    def column_type(self, f):
        p, t = C.c_void_p(), L.RuntimeType()
        L.check(self.ctx.lib.msc_dataview_column(self._h, f, C.byref(p), C.byref(t)))
        return t.type, t.count, p.value

    def column_to_numpy(self, f):
        """Device column -> numpy (tests / debugging)."""
        t, cnt, ptr = self.column_type(f)
        npt = np.dtype(_NP_OF_TYPE[t])
        nbytes = self.nrows * cnt * npt.itemsize
        self.ctx.synchronize()
        if nbytes == 0:
            return np.zeros((0, cnt) if cnt > 1 else (0,), dtype=npt)
        buf = _alias_tensor(ptr, nbytes, torch.uint8, self.ctx.torch_device)
        out = buf.cpu().numpy().view(npt)
        return out.reshape(self.nrows, cnt) if cnt > 1 else out

    def __len__(self):
        return self.nrows

    def size(self):
        return self.nrows

    def invalidate(self):
        """the adopted tensors were rewritten in place: drop what the library derived from them (msc_dataview_invalidate)"""
        L.check(self.ctx.lib.msc_dataview_invalidate(self._h))

    def close(self):
        if getattr(self, "_h", None):
            # (a view that outlives its context -- kept alive by a traceback, say -- must not hand the library a handle whose
            # context is gone: the context's destruction already released the device)
            if getattr(self.ctx, "_h", None):
                self.ctx.lib.msc_dataview_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ss_dtype(family, dim=0):
    """numpy record of one group's suff-stats as msc_state_set_ss / get_ss exchange it."""
    if family == L.BB:
        return np.dtype([("heads", np.uint32), ("tails", np.uint32)])
    if family == L.BBNC:
        return np.dtype([("heads", np.uint32), ("tails", np.uint32), ("p", np.float32)])
    if family == L.GP:
        return np.dtype([("count", np.uint32), ("sum", np.uint32), ("log_prod", np.float32)])
    if family == L.DD:
        return np.dtype([("count_sum", np.uint32), ("counts", np.uint32, (dim,))])
    if family == L.BNB:
        return np.dtype([("count", np.uint32), ("sum", np.uint32)])
    if family == L.DM:
        return np.dtype([("counts", np.uint32, (dim,)), ("ratio", np.float32)])
    if family == L.NICH:
        return np.dtype([("count", np.uint32), ("mean", np.float32), ("count_times_variance", np.float32)])
    if family == L.NIW:
        return np.dtype([("count", np.uint32), ("sum_x", np.float32, (dim,)),
                         ("sum_xxT", np.float32, (dim, dim))])
    return np.dtype([("unused", np.uint32)])


def pack_hp(family, hp, dim=0):
    """dict keyed as microscopes/models.pyx:185-290 -> the flat float block of the ABI."""
    if isinstance(hp, np.ndarray):
        return np.ascontiguousarray(hp, dtype=np.float32)
    if family in (L.BB, L.BBNC):
        v = [hp["alpha"], hp["beta"]]
    elif family == L.GP:
        v = [hp["alpha"], hp["inv_beta"]]
    elif family in (L.DD, L.DM):
        v = list(hp["alphas"])
    elif family == L.BNB:
        v = [hp["alpha"], hp["beta"], hp["r"]]
    elif family == L.NICH:
        v = [hp["mu"], hp["kappa"], hp["sigmasq"], hp["nu"]]
    elif family == L.NIW:
        v = [hp["kappa"], hp["nu"]] + list(np.asarray(hp["mu"]).ravel()) + list(np.asarray(hp["psi"]).ravel())
    else:
        v = []
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


class State(object):
    """hypers + K groups of suff-stats per feature + CRP counts, resident in HBM."""

    def __init__(self, ctx, features, ngroups):
        """features: list of (family, dim) or model descriptors with .family/.dim."""
        self.ctx = ctx
        feats = []
        for f in features:
            if isinstance(f, tuple):
                feats.append((int(f[0]), int(f[1])))
            else:
                feats.append((int(f.family), int(f.dim)))
        self.features, self.K = feats, int(ngroups)
        spec = (L.FeatureSpec * len(feats))(*[L.FeatureSpec(a, b) for a, b in feats])
        h = C.c_void_p()
        L.check(ctx.lib.msc_state_create(ctx._h, spec, len(feats), self.K, C.byref(h)))
        self._h = h

    # hypers -----------------------------------------------------------------
    def set_hp(self, f, hp):
        fam, dim = self.features[f]
        a = pack_hp(fam, hp, dim)
        L.check(self.ctx.lib.msc_state_set_hp(self._h, f, a.ctypes.data_as(C.c_void_p), a.size))

    def get_hp(self, f):
        fam, dim = self.features[f]
        a = np.empty(self.ctx.lib.msc_hp_floats(fam, dim), dtype=np.float32)
        L.check(self.ctx.lib.msc_state_get_hp(self._h, f, a.ctypes.data_as(C.c_void_p), a.size))
        return a

    # suff-stats -------------------------------------------------------------
    def set_ss(self, f, records, first_group=0):
        fam, dim = self.features[f]
        r = np.ascontiguousarray(records, dtype=ss_dtype(fam, dim))
        L.check(self.ctx.lib.msc_state_set_ss(self._h, f, first_group, r.shape[0],
                                              r.ctypes.data_as(C.c_void_p), r.nbytes))

    def get_ss(self, f, first_group=0, ngroups=None):
        fam, dim = self.features[f]
        n = self.K - first_group if ngroups is None else ngroups
        r = np.zeros(n, dtype=ss_dtype(fam, dim))
        L.check(self.ctx.lib.msc_state_get_ss(self._h, f, first_group, n, r.ctypes.data_as(C.c_void_p), r.nbytes))
        return r

    def set_alpha(self, alpha):
        L.check(self.ctx.lib.msc_state_set_alpha(self._h, float(alpha)))

    def set_group_counts(self, counts):
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        L.check(self.ctx.lib.msc_state_set_group_counts(self._h, c.ctypes.data_as(C.c_void_p), c.size))

    def get_group_counts(self):
        c = np.zeros(self.K, dtype=np.uint32)
        L.check(self.ctx.lib.msc_state_get_group_counts(self._h, c.ctypes.data_as(C.c_void_p), c.size))
        return c

    # hot path ---------------------------------------------------------------
    def _cols(self, cols):
        if cols is None:
            return None
        return (C.c_uint32 * len(cols))(*[int(c) for c in cols])

    def score_value(self, view, out=None, row0=0, nrows=None, z=None, crp_prior=False, cols=None,
                    niw_f32=False):
        """[nrows, K] float32 device tensor of summed score_value (see msc_score_value)."""
        self._bound_view = view          # (the library keeps no reference to a view: this object does, for the last one bound)
        n = view.nrows - row0 if nrows is None else nrows
        if out is None:
            out = torch.empty((n, self.K), dtype=torch.float32, device=self.ctx.torch_device)
        if out.dtype != torch.float32 or out.stride(-1) != 1 or out.shape[0] < n:
            raise ValueError("out must be a row-major float32 [nrows, >=K] tensor")
        ld = out.stride(0) if out.dim() == 2 else self.K
        zp = None
        if z is not None:
            if z.dtype != torch.int32 or not z.is_contiguous() or z.shape[0] < n:
                raise ValueError("z must be a contiguous int32 tensor of nrows entries")
            zp = C.c_void_p(z.data_ptr())
        L.check(self.ctx.lib.msc_score_value(self._h, view._h, self._cols(cols), row0, n, zp,
                                             (L.SCORE_CRP_PRIOR if crp_prior else 0) |
                                             (L.SCORE_NIW_F32 if niw_f32 else 0),
                                             C.c_void_p(out.data_ptr()), ld))
        return out

    def score_tune(self, view, out, row0=0, nrows=None, cols=None):
        """Settle the single-nich pass's launch shape for passes like this one (synchronous, ~10 ms; msc_score_tune).
        -> (shape index or -1, ms per pass)."""
        self._bound_view = view          # (the library keeps no reference to a view: this object does, for the last one bound)
        n = view.nrows - row0 if nrows is None else nrows
        if out.dtype != torch.float32 or out.stride(-1) != 1 or out.shape[0] < n:
            raise ValueError("out must be a row-major float32 [nrows, >=K] tensor")
        shape, ms = C.c_int(-1), C.c_float(0)
        L.check(self.ctx.lib.msc_score_tune(self._h, view._h, self._cols(cols), row0, n, C.c_void_p(out.data_ptr()),
                                            out.stride(0) if out.dim() == 2 else self.K, C.byref(shape), C.byref(ms)))
        return shape.value, ms.value

    def accumulate(self, view, z, row0=0, nrows=None, reset=True, subtract=False, commit=True, cols=None):
        self._bound_view = view          # (the library keeps no reference to a view: this object does, for the last one bound)
        n = view.nrows - row0 if nrows is None else nrows
        if z.dtype != torch.int32 or not z.is_contiguous() or z.shape[0] < n:
            raise ValueError("z must be a contiguous int32 tensor of nrows entries")
        flags = (L.ACC_RESET if reset else 0) | (L.ACC_SUBTRACT if subtract else 0) | \
                (0 if commit else L.ACC_NO_COMMIT)
        L.check(self.ctx.lib.msc_accumulate(self._h, view._h, self._cols(cols), row0, n,
                                            C.c_void_p(z.data_ptr()), flags))

    def entity_op(self, view, row, group, join=True, z=None, cols=None):
        """one entity joins / leaves one group, the group by value (msc_entity_op): every table stays current"""
        self._bound_view = view          # (the library keeps no reference to a view: this object does, for the last one bound)
        zp = None
        if z is not None:
            if z.dtype != torch.int32 or not z.is_contiguous() or z.shape[0] <= row:
                raise ValueError("z must be a contiguous int32 tensor covering the row")
            zp = C.c_void_p(z.data_ptr())
        L.check(self.ctx.lib.msc_entity_op(self._h, view._h, self._cols(cols), int(row), int(group), 1 if join else -1, zp))

    def score_data(self, out=None):
        if out is None:
            out = torch.empty((len(self.features), self.K), dtype=torch.float32, device=self.ctx.torch_device)
        L.check(self.ctx.lib.msc_score_data(self._h, C.c_void_p(out.data_ptr())))
        return out

    def sweep_assign(self, view, z, seed, sweep, row0=0, nrows=None, row_id0=None, cols=None):
        self._bound_view = view          # (the library keeps no reference to a view: this object does, for the last one bound)
        n = view.nrows - row0 if nrows is None else nrows
        if z.dtype != torch.int32 or not z.is_contiguous() or z.shape[0] < n:
            raise ValueError("z must be a contiguous int32 tensor of nrows entries")
        L.check(self.ctx.lib.msc_sweep_assign(self._h, view._h, self._cols(cols), row0, n,
                                              row0 if row_id0 is None else row_id0,
                                              C.c_void_p(z.data_ptr()), int(seed), int(sweep)))

    def sweep_step(self, view, z, seed, sweep, row0=0, nrows=None, row_id0=None, cols=None):
        """sweep_assign + accumulate(reset) in one call; repeated steps replay as a HIP graph (msc_sweep_step)."""
        self._bound_view = view          # (the library keeps no reference to a view: this object does, for the last one bound)
        n = view.nrows - row0 if nrows is None else nrows
        if z.dtype != torch.int32 or not z.is_contiguous() or z.shape[0] < n:
            raise ValueError("z must be a contiguous int32 tensor of nrows entries")
        L.check(self.ctx.lib.msc_sweep_step(self._h, view._h, self._cols(cols), row0, n,
                                            row0 if row_id0 is None else row_id0,
                                            C.c_void_p(z.data_ptr()), int(seed), int(sweep)))

    def sweep_step_begin(self, view, z, seed, sweep, row0=0, nrows=None, row_id0=None, cols=None):
        """the sharded step up to the exchange: follow with all-reduce of reduce_buffers() and commit_reduce()"""
        self._bound_view = view          # (the library keeps no reference to a view: this object does, for the last one bound)
        n = view.nrows - row0 if nrows is None else nrows
        if z.dtype != torch.int32 or not z.is_contiguous() or z.shape[0] < n:
            raise ValueError("z must be a contiguous int32 tensor of nrows entries")
        L.check(self.ctx.lib.msc_sweep_step_begin(self._h, view._h, self._cols(cols), row0, n,
                                                  row0 if row_id0 is None else row_id0,
                                                  C.c_void_p(z.data_ptr()), int(seed), int(sweep)))

    def sweep_step_stats(self):
        """(steps run launch by launch, steps run as one graph launch)"""
        e, g = C.c_uint64(), C.c_uint64()
        L.check(self.ctx.lib.msc_sweep_step_stats(self._h, C.byref(e), C.byref(g)))
        return e.value, g.value

    # multi-GPU hook ---------------------------------------------------------
    def reduce_buffers(self):
        """(int64 tensor, float64 tensor) aliasing the additive tables, for all_reduce(SUM)."""
        pi, ni, pf, nf = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        L.check(self.ctx.lib.msc_state_reduce_buffers(self._h, C.byref(pi), C.byref(ni), C.byref(pf), C.byref(nf)))
        return (_alias_tensor(pi.value, ni.value, torch.int64, self.ctx.torch_device),
                _alias_tensor(pf.value, nf.value, torch.float64, self.ctx.torch_device))

    def commit_reduce(self):
        L.check(self.ctx.lib.msc_state_commit_reduce(self._h))

    def reduce_pack(self):
        """both additive tables as ONE float64 tensor (a copy the state owns; one launch): all_reduce it in place, then
        reduce_unpack() and commit_reduce()"""
        p, n = C.c_void_p(), C.c_size_t()
        L.check(self.ctx.lib.msc_state_reduce_pack(self._h, C.byref(p), C.byref(n)))
        return _alias_tensor(p.value, n.value, torch.float64, self.ctx.torch_device)

    def reduce_unpack(self):
        L.check(self.ctx.lib.msc_state_reduce_unpack(self._h))

    def set_sweep_rows(self, global_rows):
        """rows of the WHOLE dataset when this state sweeps a shard through a view of its own (msc_state_set_sweep_rows)"""
        L.check(self.ctx.lib.msc_state_set_sweep_rows(self._h, int(global_rows)))

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):           # (see DataView.close)
                self.ctx.lib.msc_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DeviceBuffer(object):
    """owner of a msc_device_alloc* buffer: freed when the last tensor aliasing it is gone"""

    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr
        ctx._buffers.add(self)

    def __del__(self):
        try:
            if self.ptr and getattr(self.ctx, "_h", None):
                self.ctx.lib.msc_device_free(self.ctx._h, C.c_void_p(self.ptr))
        except Exception:
            pass
        self.ptr = None


class _CudaArrayView(object):
    def __init__(self, ptr, n, typestr, owner=None):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}
        self._owner = owner          # (torch keeps this object alive as long as the tensor's storage)


def _alias_tensor(ptr, n, dtype, device, owner=None):
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    typestr = {torch.int64: "<i8", torch.float64: "<f8", torch.uint8: "|u1", torch.float32: "<f4",
               torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_CudaArrayView(ptr, n, typestr, owner), device=device)


def _check_slice_args(scores, dim, ngroups, nd):
    """a slice reduction indexes score columns up to prod(ngroups): the matrix must have them all (the kernel skips and
    reports an offset past the row, MSC_EDEVICE, but a consistent caller never gets there)"""
    if len(ngroups) != nd or not 0 <= int(dim) < nd:
        raise ValueError("one cluster count per dimension and a dimension inside the relation")
    nblocks = 1
    for k in ngroups:
        nblocks *= int(k)
    if scores.dim() != 2 or scores.shape[1] < nblocks:
        raise ValueError("scores has %d columns but the cluster counts %s make %d blocks" %
                         (scores.shape[1] if scores.dim() == 2 else -1, list(ngroups), nblocks))


class RelationView(object):
    """A relation (numpy N-d array, optionally masked) on the device: a one-feature DataView whose rows
    are the relation's cells in row-major order (microscopes/common/relation/dataview.pyx numpy_dataview).
    `blocks(zs, ngroups)` maps per-dimension cluster assignments to the cell's block (= group) index."""

    def __init__(self, ctx, array):
        if array is None or array.ndim < 1:
            raise ValueError("need an N-d array")
        if any(int(d) == 0 for d in array.shape):
            raise ValueError("empty dims not allowed")              # relation/_dataview.pyx:33-34
        self.ctx = ctx
        self.shape = tuple(int(s) for s in array.shape)
        data = np.ascontiguousarray(np.ma.getdata(array)).reshape(-1)
        rec = np.zeros(data.shape[0], dtype=[("f0", data.dtype)])
        rec["f0"] = data
        if hasattr(array, "mask"):
            m = np.zeros(data.shape[0], dtype=[("f0", np.bool_)])
            m["f0"] = np.ascontiguousarray(np.ma.getmaskarray(array)).reshape(-1)
            rec = np.ma.masked_array(rec, mask=m)
        self.cells = DataView.from_recarray(ctx, rec)

    def blocks(self, zs, ngroups):
        """zs: one int32 device tensor per dimension; ngroups: clusters per dimension -> int32 [ncells]"""
        if len(zs) != len(self.shape) or len(ngroups) != len(self.shape):
            raise ValueError("one assignment vector and one cluster count per dimension")
        for z, n in zip(zs, self.shape):
            if z.dtype != torch.int32 or z.numel() != n or not z.is_contiguous():
                raise ValueError("assignments must be contiguous int32 tensors of the dimension's length")
        nd = len(self.shape)
        out = torch.empty(self.cells.nrows, dtype=torch.int32, device=self.ctx.torch_device)
        shape = (C.c_uint64 * nd)(*self.shape)
        zp = (C.c_void_p * nd)(*[z.data_ptr() for z in zs])
        kg = (C.c_uint32 * nd)(*[int(k) for k in ngroups])
        L.check(self.ctx.lib.msc_relation_blocks(self.ctx._h, nd, shape, zp, kg, None, self.cells.nrows,
                                                 C.c_void_p(out.data_ptr())))
        return out

    def slice_offsets(self, zs, ngroups, dim):
        """block index of every cell with dimension `dim`'s cluster taken as 0 (-1 where another entity of the cell is
        unassigned): the `off` of slice_scores.  zs[dim] is ignored."""
        zs = list(zs)
        zs[dim] = torch.zeros(self.shape[dim], dtype=torch.int32, device=self.ctx.torch_device)
        return self.blocks(zs, ngroups)

    def slice_scores(self, scores, off, dim, ngroups):
        """irm's slice reduction (msc_relation_slice_scores): out[e, g] = sum over the cells of slice (dim, e) of the
        cell's score against block (g, the cell's other clusters).  scores: [ncells, >= prod(ngroups)] from
        State.score_value on self.cells; off: slice_offsets(...).  -> float32 [shape[dim], ngroups[dim]]"""
        nd = len(self.shape)
        _check_slice_args(scores, dim, ngroups, nd)
        if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1 or scores.shape[0] != self.cells.nrows:
            raise ValueError("scores must be a row-major float32 [ncells, nblocks] tensor")
        if off.dtype != torch.int32 or off.numel() != self.cells.nrows or not off.is_contiguous():
            raise ValueError("off must be a contiguous int32 tensor of ncells entries")
        stride = 1
        for k in ngroups[dim + 1:]:
            stride *= int(k)
        out = torch.empty((self.shape[dim], int(ngroups[dim])), dtype=torch.float32, device=self.ctx.torch_device)
        shape = (C.c_uint64 * nd)(*self.shape)
        L.check(self.ctx.lib.msc_relation_slice_scores(self.ctx._h, C.c_void_p(scores.data_ptr()), scores.stride(0), nd, shape, dim,
                                                       None, None, C.c_void_p(off.data_ptr()), int(ngroups[dim]), stride,
                                                       self.shape[dim], C.c_void_p(out.data_ptr()), out.stride(0)))
        return out


class SparseRelationView(object):
    """A sparse 2-D relation (scipy.sparse matrix) on the device -- microscopes/common/relation/dataview.pyx
    sparse_2d_dataview over compressed_2darray (relation/dataview.hpp:420-578): only the stored entries are cells, in CSR
    order; their (row, column) positions travel with them.  Same operations as RelationView."""

    def __init__(self, ctx, rep):
        rows, cols = rep.shape
        if rows <= 0 or cols <= 0:
            raise ValueError("both dimensions must be positive")
        csr = rep.tocsr()
        csr.sort_indices()
        self.ctx = ctx
        self.shape = (int(rows), int(cols))
        self._csr = csr
        nnz = int(csr.nnz)
        rec = np.zeros(nnz, dtype=[("f0", csr.data.dtype)])
        rec["f0"] = csr.data
        self.cells = DataView.from_recarray(ctx, rec) if nnz else None
        row_of = np.repeat(np.arange(rows, dtype=np.uint32), np.diff(csr.indptr))
        pos = np.stack([row_of, csr.indices.astype(np.uint32)], axis=1)
        dev = ctx.torch_device
        self._pos = torch.from_numpy(np.ascontiguousarray(pos).view(np.int32).reshape(-1).copy()).to(dev)   # uint32 pairs
        # slices: rows of the CSR (dimension 0) and of its transpose (dimension 1), as cell ids
        order = np.argsort(csr.indices, kind="stable").astype(np.int32)
        colptr = np.concatenate([[0], np.cumsum(np.bincount(csr.indices, minlength=cols))]).astype(np.int32)
        self._seg = [torch.from_numpy(csr.indptr.astype(np.int32).copy()).to(dev), torch.from_numpy(colptr).to(dev)]
        self._ids = [torch.arange(nnz, dtype=torch.int32, device=dev), torch.from_numpy(order).to(dev)]

    def tocsr(self):
        return self._csr

    def nnz(self):
        return int(self._csr.nnz)

    def blocks(self, zs, ngroups):
        if len(zs) != 2 or len(ngroups) != 2:
            raise ValueError("one assignment vector and one cluster count per dimension")
        for z, n in zip(zs, self.shape):
            if z.dtype != torch.int32 or z.numel() != n or not z.is_contiguous():
                raise ValueError("assignments must be contiguous int32 tensors of the dimension's length")
        out = torch.empty(max(self.nnz(), 1), dtype=torch.int32, device=self.ctx.torch_device)
        shape = (C.c_uint64 * 2)(*self.shape)
        zp = (C.c_void_p * 2)(*[z.data_ptr() for z in zs])
        kg = (C.c_uint32 * 2)(*[int(k) for k in ngroups])
        L.check(self.ctx.lib.msc_relation_blocks(self.ctx._h, 2, shape, zp, kg, C.c_void_p(self._pos.data_ptr()), self.nnz(),
                                                 C.c_void_p(out.data_ptr())))
        return out[:self.nnz()]

    def slice_offsets(self, zs, ngroups, dim):
        zs = list(zs)
        zs[dim] = torch.zeros(self.shape[dim], dtype=torch.int32, device=self.ctx.torch_device)
        return self.blocks(zs, ngroups)

    def slice_scores(self, scores, off, dim, ngroups):
        """out[e, g] = sum over the stored cells of row / column e of the cell's score against block (g, the other
        entity's cluster) -- msc_relation_slice_scores with the CSR rows (dim 0) or the transpose's (dim 1)"""
        _check_slice_args(scores, dim, ngroups, 2)
        if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1 or scores.shape[0] != self.nnz():
            raise ValueError("scores must be a row-major float32 [nnz, nblocks] tensor")
        if off.dtype != torch.int32 or off.numel() != self.nnz() or not off.is_contiguous():
            raise ValueError("off must be a contiguous int32 tensor of nnz entries")
        stride = int(ngroups[1]) if dim == 0 else 1
        out = torch.empty((self.shape[dim], int(ngroups[dim])), dtype=torch.float32, device=self.ctx.torch_device)
        shape = (C.c_uint64 * 2)(*self.shape)
        L.check(self.ctx.lib.msc_relation_slice_scores(
            self.ctx._h, C.c_void_p(scores.data_ptr()), scores.stride(0), 2, shape, dim, C.c_void_p(self._seg[dim].data_ptr()),
            C.c_void_p(self._ids[dim].data_ptr()), C.c_void_p(off.data_ptr()), int(ngroups[dim]), stride, self.shape[dim],
            C.c_void_p(out.data_ptr()), out.stride(0)))
        return out
