"""ctypes front-end of the CPU oracle (oracle/msc_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by common_amd/.  PARITY UNPINNED against the
real reference (see oracle/msc_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BB, GP, DD, NICH, NIW, NOOP, BBNC, BNB, DM = range(9)
FAMILY_NAMES = {BB: "bb", GP: "gp", DD: "dd", NICH: "nich", NIW: "niw", NOOP: "noop", BBNC: "bbnc",
                BNB: "bnb", DM: "dm"}

(TYPE_B, TYPE_I8, TYPE_U8, TYPE_I16, TYPE_U16, TYPE_I32, TYPE_U32, TYPE_I64, TYPE_U64,
 TYPE_F32, TYPE_F64) = range(11)
NP_OF_TYPE = {TYPE_B: np.bool_, TYPE_I8: np.int8, TYPE_U8: np.uint8, TYPE_I16: np.int16,
              TYPE_U16: np.uint16, TYPE_I32: np.int32, TYPE_U32: np.uint32, TYPE_I64: np.int64,
              TYPE_U64: np.uint64, TYPE_F32: np.float32, TYPE_F64: np.float64}


def build(force=False):
    so = os.path.join(_HERE, "libmsc_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("msc_oracle.c", "msc_oracle_impl.inc", "msc_oracle.h")]
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmsc_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _declare(_LIB)
    return _LIB


def _declare(L):
    vp, u, sz, i32p = C.c_void_p, C.c_uint, C.c_size_t, C.c_void_p
    for name in ("orc_f32_ss_size", "orc_f64_ss_size", "orc_hp_size", "orc_value_size"):
        getattr(L, name).restype = sz
        getattr(L, name).argtypes = [C.c_int, u]
    for pfx, real in (("orc_f32", C.c_float), ("orc_f64", C.c_double)):
        g = lambda n: getattr(L, pfx + "_" + n)
        g("init").argtypes = [C.c_int, u, vp, vp]
        g("add_value").argtypes = [C.c_int, u, vp, vp, vp]
        g("remove_value").argtypes = [C.c_int, u, vp, vp, vp]
        g("score_value").argtypes = [C.c_int, u, vp, vp, vp]
        g("score_value").restype = real
        g("score_data").argtypes = [C.c_int, u, vp, vp]
        g("score_data").restype = real
        g("score_matrix").argtypes = [C.c_int, u, vp, vp, sz, vp, sz, vp]
        g("score_matrix_loo").argtypes = [C.c_int, u, vp, vp, sz, vp, i32p, sz, vp]
        g("accumulate").argtypes = [C.c_int, u, vp, vp, sz, vp, i32p, sz]
        g("score_data_all").argtypes = [C.c_int, u, vp, vp, sz, vp]
        g("scores_to_probs").argtypes = [vp, sz]
        g("pseudocount").argtypes = [C.c_uint64, C.c_float, sz]
        g("pseudocount").restype = real
        g("score_assignment").argtypes = [vp, sz, C.c_float]
        g("score_assignment").restype = real
        g("sweep").argtypes = [sz, vp, vp, vp, vp, vp, sz, C.c_float, vp, sz, C.c_uint64,
                               C.c_uint64, vp, vp]
    L.orc_sample_discrete.argtypes = [vp, sz, C.c_float]
    L.orc_sample_discrete.restype = sz
    L.orc_offsets_and_size.argtypes = [vp, vp, sz, vp, vp, vp]
    L.orc_primitive_size.argtypes = [C.c_int]
    L.orc_primitive_size.restype = sz
    L.orc_unpack_column.argtypes = [vp, sz, sz, sz, C.c_int, C.c_int, sz, vp]
    L.orc_uniform01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.orc_uniform01.restype = C.c_float
    L.orc_philox4x32_10.argtypes = [vp, vp, vp]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def real_dtype(prec):
    return np.float32 if prec == "f32" else np.float64


def ss_dtype(family, dim=0, prec="f64"):
    """numpy structured dtype matching the C record of one group's suff-stats."""
    R = real_dtype(prec)
    if family == BB:
        dt = np.dtype([("heads", np.uint32), ("tails", np.uint32)])
    elif family == BBNC:
        dt = np.dtype([("heads", np.uint32), ("tails", np.uint32), ("p", R)], align=True)
    elif family == GP:
        dt = np.dtype([("count", np.uint32), ("sum", np.uint32), ("log_prod", R)])
    elif family == DD:
        dt = np.dtype([("count_sum", np.uint32), ("counts", np.uint32, (dim,))])
    elif family == BNB:
        dt = np.dtype([("count", np.uint32), ("sum", np.uint32)])
    elif family == DM:
        rs = np.dtype(R).itemsize
        off = (4 * dim + rs - 1) // rs * rs
        dt = np.dtype({"names": ["counts", "ratio"], "formats": [(np.uint32, (dim,)), R],
                       "offsets": [0, off], "itemsize": off + rs})
    elif family == NICH:
        dt = np.dtype([("count", np.uint32), ("mean", R), ("count_times_variance", R)], align=True)
    elif family == NIW:
        rs = np.dtype(R).itemsize
        dt = np.dtype({"names": ["count", "sum_x", "sum_xxT"],
                       "formats": [np.uint32, (R, (dim,)), (R, (dim, dim))],
                       "offsets": [0, rs, rs * (1 + dim)],
                       "itemsize": rs * (1 + dim + dim * dim)})
    else:
        dt = np.dtype([("unused", np.uint32)])
    want = getattr(lib(), "orc_%s_ss_size" % prec)(family, dim)
    assert dt.itemsize == want, (family, dim, prec, dt.itemsize, want)
    return dt


def value_dtype(family, dim=0):
    return {BB: np.dtype(np.uint8), BBNC: np.dtype(np.uint8), GP: np.dtype(np.uint32), DD: np.dtype(np.int32),
            NICH: np.dtype(np.float32), NIW: np.dtype((np.float32, (dim,))),
            NOOP: np.dtype(np.uint8), BNB: np.dtype(np.uint32), DM: np.dtype((np.int32, (dim,)))}[family]


def pack_hp(family, hp, dim=0):
    """dict of hyperparameters (names as microscopes/models.pyx:185-290) -> float32 block."""
    if family in (BB, BBNC):
        v = [hp["alpha"], hp["beta"]]
    elif family == GP:
        v = [hp["alpha"], hp["inv_beta"]]
    elif family in (DD, DM):
        v = list(hp["alphas"])
    elif family == BNB:
        v = [hp["alpha"], hp["beta"], hp["r"]]
    elif family == NICH:
        v = [hp["mu"], hp["kappa"], hp["sigmasq"], hp["nu"]]
    elif family == NIW:
        v = [hp["kappa"], hp["nu"]] + list(np.asarray(hp["mu"]).ravel()) + \
            list(np.asarray(hp["psi"]).ravel())
    else:
        v = []
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
    assert a.nbytes == lib().orc_hp_size(family, dim), (family, dim, a.nbytes)
    return a


def widen_ss(family, ss32, dim=0):
    """float-state record array -> the double twin's record array (exact)."""
    out = np.zeros(ss32.shape, dtype=ss_dtype(family, dim, "f64"))
    for name in ss32.dtype.names:
        out[name] = ss32[name]
    return out


def narrow_ss(family, ss64, dim=0):
    out = np.zeros(ss64.shape, dtype=ss_dtype(family, dim, "f32"))
    for name in ss64.dtype.names:
        out[name] = ss64[name]
    return out


def _vals(family, values, dim):
    return np.ascontiguousarray(values, dtype=value_dtype(family, dim).base)


class Family(object):
    """One (family, dim, hp) triple bound to a precision."""

    def __init__(self, family, hp, dim=0, prec="f64"):
        self.family, self.dim, self.prec = family, dim, prec
        self.hp = hp if isinstance(hp, np.ndarray) else pack_hp(family, hp, dim)
        self.R = real_dtype(prec)
        self._g = lambda n: getattr(lib(), "orc_%s_%s" % (prec, n))

    def new_groups(self, K):
        ss = np.zeros(K, dtype=ss_dtype(self.family, self.dim, self.prec))
        for k in range(K):
            self._g("init")(self.family, self.dim, _p(self.hp), C.c_void_p(ss.ctypes.data + k * ss.itemsize))
        return ss

    def _grp(self, ss, k):
        return C.c_void_p(ss.ctypes.data + k * ss.itemsize)

    def add_value(self, ss, k, value):
        v = _vals(self.family, value, self.dim)
        self._g("add_value")(self.family, self.dim, _p(self.hp), self._grp(ss, k), _p(v))

    def remove_value(self, ss, k, value):
        v = _vals(self.family, value, self.dim)
        self._g("remove_value")(self.family, self.dim, _p(self.hp), self._grp(ss, k), _p(v))

    def score_value(self, ss, k, value):
        v = _vals(self.family, value, self.dim)
        return self._g("score_value")(self.family, self.dim, _p(self.hp), self._grp(ss, k), _p(v))

    def score_data(self, ss, k):
        return self._g("score_data")(self.family, self.dim, _p(self.hp), self._grp(ss, k))

    def score_matrix(self, ss, values, z=None):
        v = _vals(self.family, values, self.dim)
        N, K = v.shape[0], ss.shape[0]
        out = np.empty((N, K), dtype=self.R)
        if z is None:
            self._g("score_matrix")(self.family, self.dim, _p(self.hp), _p(ss), K, _p(v), N, _p(out))
        else:
            zz = np.ascontiguousarray(z, dtype=np.int32)
            self._g("score_matrix_loo")(self.family, self.dim, _p(self.hp), _p(ss), K, _p(v),
                                        _p(zz), N, _p(out))
        return out

    def accumulate(self, K, values, z, ss_init=None):
        v = _vals(self.family, values, self.dim)
        zz = np.ascontiguousarray(z, dtype=np.int32)
        ss = np.zeros(K, dtype=ss_dtype(self.family, self.dim, self.prec)) if ss_init is None else ss_init.copy()
        self._g("accumulate")(self.family, self.dim, _p(self.hp), _p(ss), K, _p(v), _p(zz), v.shape[0])
        return ss

    def score_data_all(self, ss):
        out = np.empty(ss.shape[0], dtype=self.R)
        self._g("score_data_all")(self.family, self.dim, _p(self.hp), _p(ss), ss.shape[0], _p(out))
        return out


def scores_to_probs(scores, prec="f64"):
    s = np.array(scores, dtype=real_dtype(prec), copy=True)
    getattr(lib(), "orc_%s_scores_to_probs" % prec)(_p(s), s.shape[0])
    return s


def sample_discrete(probs, dart):
    p = np.ascontiguousarray(probs, dtype=np.float32)
    return lib().orc_sample_discrete(_p(p), p.shape[0], float(dart))


def pseudocount(count, alpha, nempty, prec="f64"):
    return getattr(lib(), "orc_%s_pseudocount" % prec)(int(count), float(alpha), int(nempty))


def score_assignment(assignments, alpha, prec="f64"):
    a = np.ascontiguousarray(assignments, dtype=np.int64)
    return getattr(lib(), "orc_%s_score_assignment" % prec)(_p(a), a.shape[0], float(alpha))


def uniform01(seed, sweep, row):
    return lib().orc_uniform01(int(seed), int(sweep), int(row))


def philox(key, ctr):
    k = np.asarray(key, dtype=np.uint32)
    c = np.asarray(ctr, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(k), _p(c), _p(out))
    return out


def offsets_and_size(prim_types, counts):
    t = np.ascontiguousarray(prim_types, dtype=np.int32)
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    off = np.zeros(len(t), dtype=np.uint64)
    row = C.c_size_t()
    mrow = C.c_size_t()
    lib().orc_offsets_and_size(_p(t), _p(c), len(t), _p(off), C.byref(row), C.byref(mrow))
    return off.astype(np.int64), row.value, mrow.value


def unpack_column(records, rowsize, offset, elem, src_type, dst_type, n):
    rec = np.ascontiguousarray(records).view(np.uint8).ravel()
    out = np.zeros(n, dtype=NP_OF_TYPE[dst_type])
    lib().orc_unpack_column(_p(rec), rowsize, offset, elem, src_type, dst_type, n, _p(out))
    return out


def sweep(features, K, alpha, z_in, seed, sweep_idx, prec="f64", want_scores=False):
    """features: list of (Family, ss_records, values).  Returns z_out[, scores]."""
    nf = len(features)
    fam = (C.c_int * nf)(*[f.family for f, _, _ in features])
    dim = (C.c_uint * nf)(*[f.dim for f, _, _ in features])
    vals = [_vals(f.family, v, f.dim) for f, _, v in features]
    hp = (C.c_void_p * nf)(*[f.hp.ctypes.data for f, _, _ in features])
    ss = (C.c_void_p * nf)(*[s.ctypes.data for _, s, _ in features])
    vv = (C.c_void_p * nf)(*[v.ctypes.data for v in vals])
    z = np.ascontiguousarray(z_in, dtype=np.int32)
    N = z.shape[0]
    zo = np.empty(N, dtype=np.int32)
    sc = np.empty((N, K), dtype=real_dtype(prec)) if want_scores else None
    getattr(lib(), "orc_%s_sweep" % prec)(nf, fam, dim, hp, ss, vv, K, float(alpha), _p(z), N,
                                          int(seed), int(sweep_idx), _p(zo),
                                          _p(sc) if want_scores else None)
    return (zo, sc) if want_scores else zo
