"""Device-side evaluation of the BBOB f15-f24 objectives for runs that advance in lock-step.

The seeded tables of a problem (x_opt, rotations, conditioning, Gallagher peaks) come from `pcabo.bbob` - the legacy
generators are sequential integer recurrences and run once per run on the host - and are uploaded once; the B candidates
of a lock-step iteration are then evaluated in ONE launch (`pcabo_bbob_eval`, csrc/kernels_bbob.hip), which also applies
PCA_BO's out-of-box rule (reference PCA_BO.py:248-263).  Values agree with `BBOBProblem.raw` to ~1e-13 relative (other
summation order); the host evaluation stays the default of `pcabo.batchrun` so that a batched run is bit-identical to the
same run alone.  Per-run table layout (doubles): [x_opt(d) | R(d*d) | M(d*d) | aux(101*(2d+1) + 4d + 8)]."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native
from .bbob import BBOBProblem

_PEAKS = 101


def _table(p: BBOBProblem, stride: int) -> np.ndarray:
    st, d, fid = p._state, int(p.meta_data.n_variables), int(p.meta_data.problem_id)
    t = np.zeros(stride)
    xo, r, m, aux = 0, d, d + d * d, d + 2 * d * d
    put = lambda off, a: t.__setitem__(slice(off, off + np.asarray(a).size), np.asarray(a, dtype=np.float64).ravel())
    if fid in (15, 16, 17, 18, 23):
        put(xo, st.xopt); put(r, st.rot_r); put(m, st.m)
    elif fid == 19:
        put(m, st.m)
    elif fid == 20:
        put(aux, st.sign); put(aux + d, st.offset); put(aux + 2 * d, st.cond)
    elif fid in (21, 22):
        put(r, st.rot)
        P = st.peaks
        put(aux, st.heights); put(aux + _PEAKS, st.scales); put(aux + _PEAKS + _PEAKS * d, st.centres)
        assert P <= _PEAKS
    elif fid == 24:
        put(xo, st.xopt); put(r, st.rot_r); put(m, st.rot_q); put(aux, st.cond)
    else:
        raise NotImplementedError(fid)
    return t


class DeviceObjectives:
    """`problems`: B `BBOBProblem`s of one dimension.  `evaluate(X[B, d])` -> (f[B] incl. f_opt or the penalty, oob[B])."""

    def __init__(self, problems, device: int = 0, penalty: float = 1000.0):
        self.problems = list(problems)
        self.B = len(self.problems)
        self.d = int(self.problems[0].meta_data.n_variables)
        assert all(int(p.meta_data.n_variables) == self.d for p in self.problems)
        lib = _native.LIB
        lib.pcabo_bbob_table_doubles.argtypes = [C.c_int]
        lib.pcabo_bbob_table_doubles.restype = C.c_int
        lib.pcabo_bbob_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.pcabo_bbob_destroy.argtypes = [C.c_void_p]
        lib.pcabo_bbob_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        stride = lib.pcabo_bbob_table_doubles(self.d)
        if stride < 0:
            raise _native.PcaboError(stride, "dimension not supported by the device objectives")
        tables = np.stack([_table(p, stride) for p in self.problems])
        fid = np.array([int(p.meta_data.problem_id) for p in self.problems], dtype=np.int32)
        self.f_opt = np.array([p.f_opt for p in self.problems])
        self.lb, self.ub = float(self.problems[0].bounds.lb[0]), float(self.problems[0].bounds.ub[0])
        self.penalty = float(penalty)
        self._h = C.c_void_p()
        rc = lib.pcabo_bbob_create(int(device), self.B, self.d, fid.ctypes.data_as(C.c_void_p),
                                   tables.ctypes.data_as(C.c_void_p), C.byref(self._h))
        if rc != 0:
            if self._h:
                lib.pcabo_bbob_destroy(self._h)
            raise _native.PcaboError(rc, "pcabo_bbob_create failed (no usable HIP device?)")

    def evaluate(self, X: np.ndarray):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(self.B, self.d)
        raw, oob = np.empty(self.B), np.zeros(self.B, dtype=np.int32)
        rc = _native.LIB.pcabo_bbob_eval(self._h, X.ctypes.data_as(C.c_void_p), self.lb, self.ub, self.penalty,
                                         raw.ctypes.data_as(C.c_void_p), oob.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise _native.PcaboError(rc, "pcabo_bbob_eval failed")
        f = np.where(oob != 0, self.penalty, raw + self.f_opt)
        return f, raw, oob.astype(bool)

    def close(self) -> None:
        if getattr(self, "_h", None):
            _native.LIB.pcabo_bbob_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
