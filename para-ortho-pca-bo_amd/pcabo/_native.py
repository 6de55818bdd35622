"""ctypes binding of libpcabo.so (C ABI declared in include/pcabo.h).

Thin by design: numpy arrays (or torch tensors' device pointers) go in as plain pointers + sizes,
every status code is turned into a Python exception, and nothing here computes.  If the shared
library is missing the import of this module fails loudly - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

ABI_VERSION = 1
KERNEL_MATERN52, KERNEL_RBF = 0, 1
ACQ_LOG_EI, ACQ_PI = 0, 1
PTR_HOST, PTR_DEVICE = 0, 1
OPT_RESIDENT, OPT_BESTF_F32, OPT_GROUP_ACQ, OPT_DEVICE_LBFGSB, OPT_LBFGSB_CUS = 0, 1, 2, 3, 4
PROFILE_GROUPS = ("wpca", "gram", "cholesky", "root_inverse_alpha", "acq_partial", "acq_large_batches")

# Hardware queues.  A Batch drives the GPU from several worker threads on separate HIP streams (its gangs): the runtime's
# default of 4 hardware queues per process makes streams share queues, and kernels that share a queue run one after the other
# (-15 % for a 30-run batch).  The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises; this library does NOT touch the
# environment - the entry points that own the process (main.py, bench.py, tools/) set it before the first GPU call, and
# Batch() warns when it is missing or too small.
HW_QUEUES_ENV, HW_QUEUES_WANTED = "GPU_MAX_HW_QUEUES", 16


def hw_queues_advice(streams: int) -> Optional[str]:
    """None if the environment provides enough hardware queues for `streams` concurrent streams, else what to set."""
    try:
        have = int(os.environ.get(HW_QUEUES_ENV, "4"))
    except ValueError:
        have = 4
    if have >= streams:
        return None
    return (f"{HW_QUEUES_ENV}={os.environ.get(HW_QUEUES_ENV, 'unset (default 4)')}: {streams} streams of this batch will share "
            f"hardware queues and their kernels serialise; set {HW_QUEUES_ENV}={max(HW_QUEUES_WANTED, streams)} in the environment "
            "before the process makes its first GPU call")


_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lib", "libpcabo.so")

EXPORTS = [
    "pcabo_abi_version", "pcabo_device_count", "pcabo_ctx_create", "pcabo_ctx_destroy",
    "pcabo_set_pointer_mode", "pcabo_set_option", "pcabo_last_error", "pcabo_wpca", "pcabo_gp_condition", "pcabo_gp_condition_begin",
    "pcabo_gp_condition_end", "pcabo_gp_condition_end_eval", "pcabo_wpca_gp_condition_begin", "pcabo_wpca_results",
    "pcabo_acq_bounds",
    "pcabo_acq_eval", "pcabo_logei", "pcabo_optimize_acqf", "pcabo_inverse_map", "pcabo_get_gp_state",
    "pcabo_get_gram", "pcabo_lbfgsb_minimize", "pcabo_lbfgsb_set_vector_kernels", "pcabo_lbfgsb_set_sum_order", "pcabo_sobol_scramble", "pcabo_sobol_draw", "pcabo_sobol_draw_rows", "pcabo_torch_randint2", "pcabo_torch_multinomial_rows", "pcabo_boltzmann_pick_rows", "pcabo_set_profiling",
    "pcabo_get_profile", "pcabo_get_profile_calibration", "pcabo_reset_profile",
    "pcabo_batch_create", "pcabo_batch_destroy", "pcabo_batch_last_error", "pcabo_batch_ctx",
    "pcabo_batch_wpca_gp_condition_begin", "pcabo_batch_wpca_results", "pcabo_batch_acq_bounds",
    "pcabo_batch_gp_condition_end_eval", "pcabo_batch_optimize_acqf", "pcabo_batch_inverse_map", "pcabo_batch_device_acq_eval",
    "pcabo_batch_busy", "pcabo_batch_gp_condition_end_eval_begin", "pcabo_batch_gp_condition_end_eval_end",
    "pcabo_batch_optimize_acqf_begin", "pcabo_batch_optimize_acqf_end", "pcabo_batch_inverse_map_begin", "pcabo_batch_inverse_map_end",
    "pcabo_batch_set_input_strides", "pcabo_batch_gp_condition_begin",
    "pcabo_batch_set_profiling", "pcabo_batch_get_profile", "pcabo_batch_set_active", "pcabo_batch_set_workers", "pcabo_batch_set_option",
    "pcabo_device_lbfgsb_limits",
    "pcabo_bbob_table_doubles", "pcabo_bbob_create", "pcabo_bbob_destroy", "pcabo_bbob_eval",
    "pcabo_comm_unique_id", "pcabo_comm_create", "pcabo_gather_best", "pcabo_comm_last_error", "pcabo_comm_destroy",
]


class PcaboError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpcabo error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    # PCABO_LIB selects another build of the SAME library (diagnostic builds such as libpcabo_timing.so)
    return os.path.normpath(os.environ.get("PCABO_LIB") or _LIB_PATH)


def _load() -> C.CDLL:
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C para-ortho-pca-bo_amd/csrc`).  There is no CPU fallback for the HIP path.")
    lib = C.CDLL(path)
    lib.pcabo_abi_version.restype = C.c_int
    if lib.pcabo_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version mismatch")
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    lib.pcabo_device_count.restype = C.c_int
    lib.pcabo_ctx_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.pcabo_ctx_destroy.argtypes = [vp]
    lib.pcabo_set_pointer_mode.argtypes = [vp, C.c_int]
    lib.pcabo_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.pcabo_last_error.argtypes = [vp, C.c_char_p, C.c_int]
    lib.pcabo_wpca.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp,
                               vp, vp, vp, vp, ip, vp]
    lib.pcabo_gp_condition.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_double, C.c_double, C.c_int]
    lib.pcabo_gp_condition_begin.argtypes = lib.pcabo_gp_condition.argtypes
    lib.pcabo_gp_condition_end.argtypes = [vp]
    lib.pcabo_wpca_results.argtypes = [vp, vp, vp, vp, vp, ip]
    lib.pcabo_gp_condition_end_eval.argtypes = [vp, vp, C.c_int, C.c_double, C.c_int, C.c_int, vp]
    lib.pcabo_wpca_gp_condition_begin.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp,
                                                  vp, C.c_double, C.c_double, C.c_int, vp, vp, vp, vp, ip]
    lib.pcabo_acq_bounds.argtypes = [vp, vp]
    lib.pcabo_acq_eval.argtypes = [vp, vp, C.c_int, C.c_double, C.c_int, C.c_int, vp, vp]
    lib.pcabo_logei.argtypes = [vp, vp, C.c_int, C.c_double, C.c_int, vp, vp]
    lib.pcabo_optimize_acqf.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_double, C.c_int, C.c_int,
                                        vp, vp, vp, ip]
    lib.pcabo_inverse_map.argtypes = [vp, vp, vp]
    lib.pcabo_get_gp_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.pcabo_get_gram.argtypes = [vp, vp]
    lib.pcabo_sobol_scramble.argtypes = [vp, vp, C.c_int]
    lib.pcabo_sobol_draw.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.pcabo_sobol_draw_rows.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_longlong, vp]
    lib.pcabo_set_profiling.argtypes = [vp, C.c_int]
    lib.pcabo_get_profile.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_int64), dp, dp]
    lib.pcabo_reset_profile.argtypes = [vp]
    lib.pcabo_get_profile_calibration.argtypes = [vp, dp, dp]
    lib.pcabo_lbfgsb_set_vector_kernels.argtypes = [C.c_int]
    lib.pcabo_batch_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.pcabo_comm_unique_id.argtypes = [C.c_char_p]
    lib.pcabo_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.pcabo_gather_best.argtypes = [vp, vp, C.c_int, vp]
    lib.pcabo_comm_last_error.argtypes = [vp, C.c_char_p, C.c_int]
    lib.pcabo_comm_destroy.argtypes = [vp]
    lib.pcabo_batch_destroy.argtypes = [vp]
    lib.pcabo_batch_set_workers.argtypes = [vp, C.c_int]
    lib.pcabo_batch_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.pcabo_batch_last_error.argtypes = [vp, C.c_char_p, C.c_int]
    lib.pcabo_batch_ctx.argtypes = [vp, C.c_int]
    lib.pcabo_batch_wpca_gp_condition_begin.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                                        C.c_double, C.c_double, C.c_int]
    lib.pcabo_batch_wpca_results.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.pcabo_batch_acq_bounds.argtypes = [vp, vp]
    lib.pcabo_batch_gp_condition_end_eval.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    lib.pcabo_batch_optimize_acqf.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.pcabo_batch_inverse_map.argtypes = [vp, vp, vp]
    lib.pcabo_batch_device_acq_eval.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    lib.pcabo_batch_busy.argtypes = [vp]
    lib.pcabo_batch_gp_condition_end_eval_begin.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int]
    lib.pcabo_batch_gp_condition_end_eval_end.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    lib.pcabo_batch_optimize_acqf_begin.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int]
    lib.pcabo_batch_optimize_acqf_end.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.pcabo_batch_inverse_map_begin.argtypes = [vp, vp]
    lib.pcabo_batch_set_input_strides.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_size_t]
    lib.pcabo_batch_gp_condition_begin.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_double, C.c_double, C.c_int]
    lib.pcabo_batch_inverse_map_end.argtypes = [vp, vp]
    lib.pcabo_batch_set_profiling.argtypes = [vp, C.c_int]
    lib.pcabo_batch_set_active.argtypes = [vp, vp]
    lib.pcabo_batch_get_profile.argtypes = [vp, vp]
    for name in EXPORTS:
        getattr(lib, name).restype = C.c_int
    lib.pcabo_batch_ctx.restype = vp
    return lib


LIB = _load()
FG_CALLBACK = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
LIB.pcabo_lbfgsb_minimize.argtypes = [
    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, FG_CALLBACK, C.c_void_p, C.c_int, C.c_double, C.c_double,
    C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]


def device_count() -> int:
    return int(LIB.pcabo_device_count())


def device_lbfgsb_limits():
    """(max n, max k, max points per restart group) of the device-resident optimiser (pcabo_device_lbfgsb_limits; host code)."""
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    LIB.pcabo_device_lbfgsb_limits(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _ptr(a: Optional[np.ndarray]):
    # (the address as an int - ctypes converts it for a c_void_p parameter; half the cost of data_as(), and a lock-step
    # iteration of 30 runs hands ~180 arrays to the library.  The caller keeps `a` alive across the call.)
    return None if a is None else a.ctypes.data


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context:
    """One per-run device context (a HIP stream + workspaces).  Not thread-safe; use one per thread."""

    def __init__(self, max_n: int, max_d: int, max_q: int = 512, device: int = 0):
        self._h = C.c_void_p()
        rc = LIB.pcabo_ctx_create(int(device), int(max_n), int(max_d), int(max_q), C.byref(self._h))
        if rc != 0:
            msg = self._err() if self._h else "no usable HIP device (the HIP path has no CPU fallback)"
            if self._h:
                LIB.pcabo_ctx_destroy(self._h)
                self._h = C.c_void_p()
            raise PcaboError(rc, msg)
        self.max_n, self.max_d, self.max_q, self.device = max_n, max_d, max_q, device
        self.n = self.d = self.k = 0

    def _err(self) -> str:
        buf = C.create_string_buffer(512)
        LIB.pcabo_last_error(self._h, buf, 512)
        return buf.value.decode(errors="replace")

    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise PcaboError(rc, self._err())

    def set_option(self, option: int, value: int) -> None:
        self._chk(LIB.pcabo_set_option(self._h, int(option), int(value)))

    def match_best_f_dtype(self, best_f) -> None:
        """botorch stores `torch.as_tensor(best_f)`: float32 for a Python float (or int), float64 for a numpy float64
        scalar (what a callable objective may return).  Tell the library which of the two the caller's value is."""
        f32 = 0 if isinstance(best_f, np.floating) and best_f.dtype == np.float64 else 1
        if getattr(self, "_bestf_f32", 1) != f32:
            self.set_option(OPT_BESTF_F32, f32)
            self._bestf_f32 = f32

    def close(self) -> None:
        if getattr(self, "_h", None):
            LIB.pcabo_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- rows A-C -----------------------------------------------------------------------------
    def wpca(self, X, f=None, ranks=None, maximize=False, var_threshold=0.95, n_components=0, noise=None,
             want_Z=True, want_full=True):
        X = _f64(X)
        n, d = X.shape
        f_a = None if f is None else _f64(f, (n,))
        r_a = None if ranks is None else np.ascontiguousarray(ranks, dtype=np.int64).reshape(n)
        nz = None if noise is None else _f64(noise, (n, d))
        rc_ = min(n, d)
        data_mean, pca_mean = np.empty(d), np.empty(d)
        comps = np.empty((rc_, d)) if want_full else None
        evr = np.empty(rc_) if want_full else None
        k = C.c_int(0)
        Z = np.empty((n, d)) if want_Z else None
        self._chk(LIB.pcabo_wpca(self._h, _ptr(X), _ptr(f_a), _ptr(r_a), n, d, int(bool(maximize)),
                                 float(var_threshold), int(n_components), _ptr(nz), _ptr(data_mean), _ptr(pca_mean),
                                 _ptr(comps), _ptr(evr), C.byref(k), _ptr(Z)))
        self.n, self.d, self.k = n, d, k.value
        if want_Z:
            Z = Z.reshape(-1)[: n * k.value].reshape(n, k.value).copy()
        return {"data_mean": data_mean, "pca_mean": pca_mean, "components": comps, "evr": evr, "k": k.value, "Z": Z}

    # ---- rows A-H in one enqueue -------------------------------------------------------------
    def wpca_gp_condition(self, X, y, f=None, ranks=None, maximize=False, var_threshold=0.95, n_components=0,
                          noise=None, lengthscale=0.6931471805599453, gp_noise=0.006737946999085467,
                          kernel=KERNEL_MATERN52, collect=True):
        """wpca(...) + gp_condition(y, wait=False) without the host round trip between them: returns the wPCA results
        while the conditioning is still running; call gp_wait() before using the GP.  collect=False only enqueues
        (returns None); wpca_results() then waits for the wPCA and returns its results."""
        X = _f64(X)
        n, d = X.shape
        y = _f64(y, (n,))
        f_a = None if f is None else _f64(f, (n,))
        r_a = None if ranks is None else np.ascontiguousarray(ranks, dtype=np.int64).reshape(n)
        nz = None if noise is None else _f64(noise, (n, d))
        self._chk(LIB.pcabo_wpca_gp_condition_begin(
            self._h, _ptr(X), _ptr(f_a), _ptr(r_a), n, d, int(bool(maximize)), float(var_threshold),
            int(n_components), _ptr(nz), _ptr(y), float(lengthscale), float(gp_noise), int(kernel), None, None, None, None,
            None))
        self.n, self.d = n, d
        return self.wpca_results() if collect else None

    def wpca_results(self):
        n, d = self.n, self.d
        rc_ = min(n, d)
        data_mean, pca_mean, comps, evr = np.empty(d), np.empty(d), np.empty((rc_, d)), np.empty(rc_)
        k = C.c_int(0)
        self._chk(LIB.pcabo_wpca_results(self._h, _ptr(data_mean), _ptr(pca_mean), _ptr(comps), _ptr(evr), C.byref(k)))
        self.k = k.value
        return {"data_mean": data_mean, "pca_mean": pca_mean, "components": comps, "evr": evr, "k": k.value, "Z": None}

    # ---- rows D-H -----------------------------------------------------------------------------
    def gp_condition(self, y, Z=None, norm_bounds=None, lengthscale=0.6931471805599453,
                     noise=0.006737946999085467, kernel=KERNEL_MATERN52, wait=True):
        """wait=False enqueues the conditioning and returns; call gp_wait() before using the GP."""
        y = _f64(y).reshape(-1)
        n = y.shape[0]
        if Z is not None:
            Z = _f64(Z)
            k = Z.shape[1]
        else:
            k = self.k
        nb = None if norm_bounds is None else _f64(norm_bounds, (2, k))
        self._keep = (y, Z, nb)      # host buffers must outlive the asynchronous copies
        fn = LIB.pcabo_gp_condition if wait else LIB.pcabo_gp_condition_begin
        self._chk(fn(self._h, _ptr(Z), _ptr(y), n, k, _ptr(nb), float(lengthscale), float(noise), int(kernel)))
        self.n, self.k = n, k

    def gp_wait(self) -> None:
        self._chk(LIB.pcabo_gp_condition_end(self._h))
        self._keep = None

    def gp_wait_eval(self, Xq, best_f, maximize=False, acq=ACQ_LOG_EI) -> np.ndarray:
        """gp_wait() + acq_eval(Xq, grad=False) with the evaluation enqueued behind the conditioning."""
        Xq = _f64(Xq).reshape(-1, self.k)
        q = Xq.shape[0]
        val = np.empty(q)
        self._chk(LIB.pcabo_gp_condition_end_eval(self._h, _ptr(Xq), q, float(best_f), int(bool(maximize)), int(acq),
                                                  _ptr(val)))
        self._keep = None
        return val

    def acq_bounds(self) -> np.ndarray:
        b = np.empty((2, self.k))
        self._chk(LIB.pcabo_acq_bounds(self._h, _ptr(b)))
        return b

    # ---- row I --------------------------------------------------------------------------------
    def acq_eval(self, Xq, best_f, maximize=False, acq=ACQ_LOG_EI, grad=True):
        Xq = _f64(Xq).reshape(-1, self.k)
        q = Xq.shape[0]
        val = np.empty(q)
        g = np.empty((q, self.k)) if grad else None
        self._chk(LIB.pcabo_acq_eval(self._h, _ptr(Xq), q, float(best_f), int(bool(maximize)), int(acq), _ptr(val),
                                     _ptr(g)))
        return (val, g) if grad else val

    # ---- rows M-N -----------------------------------------------------------------------------
    def optimize_acqf(self, ics, bounds, best_f, maximize=False, acq=ACQ_LOG_EI, batch_limit=5, maxiter=200):
        ics = _f64(ics).reshape(-1, self.k)
        nr = ics.shape[0]
        bounds = _f64(bounds, (2, self.k))
        cand = np.empty((nr, self.k))
        vals = np.empty(nr)
        ng = (nr + batch_limit - 1) // batch_limit
        info = np.zeros((ng, 4), dtype=np.int32)
        failed = C.c_int(0)
        self._chk(LIB.pcabo_optimize_acqf(self._h, _ptr(ics), nr, int(batch_limit), _ptr(bounds), int(maxiter),
                                          float(best_f), int(bool(maximize)), int(acq), _ptr(cand), _ptr(vals),
                                          _ptr(info), C.byref(failed)))
        return cand, vals, info, bool(failed.value)

    # ---- row O --------------------------------------------------------------------------------
    def inverse_map(self, z) -> np.ndarray:
        z = _f64(z).reshape(-1)
        x = np.empty(self.d)
        self._chk(LIB.pcabo_inverse_map(self._h, _ptr(z), _ptr(x)))
        return x

    # ---- introspection / profiling ------------------------------------------------------------
    def gp_state(self):
        n, k = self.n, self.k
        L, R, alpha, ys, nb = np.empty((n, n)), np.empty((n, n)), np.empty(n), np.empty(2), np.empty((2, k))
        self._chk(LIB.pcabo_get_gp_state(self._h, _ptr(L), _ptr(R), _ptr(alpha), _ptr(ys), _ptr(nb)))
        return {"L": np.tril(L), "R": np.tril(R), "alpha": alpha, "y_mean": ys[0], "y_std": ys[1], "norm_bounds": nb}

    def gram(self) -> np.ndarray:
        K = np.empty((self.n, self.n))
        self._chk(LIB.pcabo_get_gram(self._h, _ptr(K)))
        return K

    def set_profiling(self, on: bool) -> None:
        self._chk(LIB.pcabo_set_profiling(self._h, int(bool(on))))

    def reset_profile(self) -> None:
        self._chk(LIB.pcabo_reset_profile(self._h))

    def profile(self) -> dict:
        out = {}
        for i, name in enumerate(PROFILE_GROUPS):
            ms, cnt, by, fl = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_double(0)
            self._chk(LIB.pcabo_get_profile(self._h, i, C.byref(ms), C.byref(cnt), C.byref(by), C.byref(fl)))
            out[name] = {"ms": ms.value, "launches": cnt.value, "bytes": by.value, "flops": fl.value}
        return out

    def profile_calibration(self) -> dict:
        """Milliseconds an event pair reads with nothing in between (subtracted from every profiled pair) and around an
        empty kernel - medians of 64, taken when profiling was switched on."""
        a, b = C.c_double(0), C.c_double(0)
        self._chk(LIB.pcabo_get_profile_calibration(self._h, C.byref(a), C.byref(b)))
        return {"pair_ms": a.value, "empty_kernel_ms": b.value}


class _BorrowedContext(Context):
    """Run b's context inside a Batch: every single-context call works on it; the batch owns and frees it."""

    def __init__(self, handle, max_n, max_d, max_q, device):   # noqa: D401 - no pcabo_ctx_create here
        self._h = C.c_void_p(handle)
        self.max_n, self.max_d, self.max_q, self.device = max_n, max_d, max_q, device
        self.n = self.d = self.k = 0

    def close(self) -> None:
        self._h = C.c_void_p()


class Batch:
    """B per-run contexts advancing in lock-step (pcabo_batch_* of include/pcabo.h): one launch sequence for the
    rows A-H of all runs, one scoring launch, shared acquisition launches for the L-BFGS-B rounds of all runs."""

    def __init__(self, B: int, max_n: int, max_d: int, max_q: int = 512, device: int = 0, workers: int = 0,
                 group_acq: bool = True, device_lbfgsb: int = 0, lbfgsb_cus: int = 0):
        self._h = C.c_void_p()
        rc = LIB.pcabo_batch_create(int(device), int(B), int(max_n), int(max_d), int(max_q), C.byref(self._h))
        if rc != 0:
            msg = self._err() if self._h else "no usable HIP device (the HIP path has no CPU fallback)"
            if self._h:
                LIB.pcabo_batch_destroy(self._h)
                self._h = C.c_void_p()
            raise PcaboError(rc, msg)
        self.B, self.max_n, self.max_d, self.max_q, self.device = B, max_n, max_d, max_q, device
        self.n = self.d = 0
        self.k = np.zeros(B, dtype=np.int32)
        self.ctx = [_BorrowedContext(LIB.pcabo_batch_ctx(self._h, b), max_n, max_d, max_q, device) for b in range(B)]
        if workers:
            self.set_workers(workers)
        if not group_acq:            # L-BFGS-B rounds through the per-query kernels (a stand-alone context's default)
            self._chk(LIB.pcabo_batch_set_option(self._h, OPT_GROUP_ACQ, 0))
        if device_lbfgsb:            # 1: every restart group's L-BFGS-B inside one launch; 2: its host-stepped twin (tests)
            self._chk(LIB.pcabo_batch_set_option(self._h, OPT_DEVICE_LBFGSB, int(device_lbfgsb)))
        self.device_lbfgsb = int(device_lbfgsb)
        if lbfgsb_cus:               # the optimiser's launches confined to that many CUs (several batches of one process in flight)
            self._chk(LIB.pcabo_batch_set_option(self._h, OPT_LBFGSB_CUS, int(lbfgsb_cus)))
        advice = hw_queues_advice(min(B, int(workers) if workers else 8) + 2)      # gang streams + the batch's + the default stream
        if advice:
            import warnings
            warnings.warn(advice, RuntimeWarning, stacklevel=2)

    def set_workers(self, workers: int) -> None:
        """Worker threads of the L-BFGS-B phase (they spin: several batches of one process share the process's cores)."""
        self._chk(LIB.pcabo_batch_set_workers(self._h, int(workers)))

    def _err(self) -> str:
        buf = C.create_string_buffer(512)
        LIB.pcabo_batch_last_error(self._h, buf, 512)
        return buf.value.decode(errors="replace")

    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise PcaboError(rc, self._err())

    def close(self) -> None:
        if getattr(self, "_h", None):
            for c in self.ctx:
                c.close()
            LIB.pcabo_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def wpca_gp_condition_begin(self, X, ranks, noise, y, maximize=False, var_threshold=0.95, n_components=0,
                                lengthscale=0.6931471805599453, gp_noise=0.006737946999085467, kernel=KERNEL_MATERN52):
        """X[B,n,d], ranks[B,n] (int64), noise[B,n,d] or None, y[B,n]: enqueue rows A-H of every run."""
        X = np.asarray(X, dtype=np.float64)
        B, n, d = X.shape
        assert B == self.B
        ranks = np.ascontiguousarray(ranks, dtype=np.int64).reshape(B, n)
        y = np.asarray(y, dtype=np.float64).reshape(B, n)

        def block(a, row):          # the runs' blocks as they lie in the caller's array: a stride instead of a dense copy
            if a is None:
                return None, 0
            a = np.asarray(a, dtype=np.float64)
            item = a.itemsize
            inner_ok = all(a.strides[i] == item * int(np.prod(a.shape[i + 1:])) for i in range(1, a.ndim))
            if a.dtype == np.float64 and inner_ok and a.strides[0] % item == 0 and a.strides[0] // item >= row:
                return a, a.strides[0] // item
            a = np.ascontiguousarray(a, dtype=np.float64)
            return a, row
        X, sx = block(X, n * d)
        nz, sn = block(noise, n * d)
        y, sy = block(y, n)
        strides = (0 if sx == n * d else sx, 0 if sn == n * d else sn, 0 if sy == n else sy)
        if strides != getattr(self, "_in_strides", (0, 0, 0)):
            self._chk(LIB.pcabo_batch_set_input_strides(self._h, *strides))
            self._in_strides = strides
        self._chk(LIB.pcabo_batch_wpca_gp_condition_begin(self._h, _ptr(X), _ptr(ranks), _ptr(nz), _ptr(y), n, d,
                                                          int(bool(maximize)), float(var_threshold), int(n_components),
                                                          float(lengthscale), float(gp_noise), int(kernel)))
        self.n, self.d = n, d

    def gp_condition_begin(self, Z, y, norm_bounds=None, lengthscale=0.6931471805599453, gp_noise=0.006737946999085467,
                           kernel=KERNEL_MATERN52):
        """Z[B,n,k], y[B,n]: rows D-H of every run on the given points, no weighted PCA (Vanilla_BO); norm_bounds (2, k) for all
        runs or None.  Finish with gp_wait_eval / gp_eval_begin + gp_eval_end."""
        Z = np.asarray(Z, dtype=np.float64)
        B, n, k = Z.shape
        assert B == self.B
        y = np.asarray(y, dtype=np.float64).reshape(B, n)

        def block(a, row):
            item = a.itemsize
            inner_ok = all(a.strides[i] == item * int(np.prod(a.shape[i + 1:])) for i in range(1, a.ndim))
            if inner_ok and a.strides[0] % item == 0 and a.strides[0] // item >= row:
                return a, a.strides[0] // item
            return np.ascontiguousarray(a), row
        Z, sx = block(Z, n * k)
        y, sy = block(y, n)
        strides = (0 if sx == n * k else sx, 0, 0 if sy == n else sy)
        if strides != getattr(self, "_in_strides", (0, 0, 0)):
            self._chk(LIB.pcabo_batch_set_input_strides(self._h, *strides))
            self._in_strides = strides
        nb = None if norm_bounds is None else _f64(norm_bounds, (2, k))
        self._chk(LIB.pcabo_batch_gp_condition_begin(self._h, _ptr(Z), _ptr(y), n, k, _ptr(nb), float(lengthscale), float(gp_noise),
                                                     int(kernel)))
        self.n, self.d = n, k
        self.k = np.full(B, k, dtype=np.int32)
        for c in self.ctx:
            c.n, c.d, c.k = n, k, k

    def wpca_results(self):
        B, n, d = self.B, self.n, self.d
        rc_ = min(n, d)
        dm, pm, comps, evr = np.empty((B, d)), np.empty((B, d)), np.empty((B, d, d)), np.empty((B, d))
        k = np.zeros(B, dtype=np.int32)
        self._chk(LIB.pcabo_batch_wpca_results(self._h, _ptr(dm), _ptr(pm), _ptr(comps), _ptr(evr), _ptr(k)))
        self.k = k
        for b, c in enumerate(self.ctx):
            c.n, c.d, c.k = n, d, int(k[b])
        return [{"data_mean": dm[b], "pca_mean": pm[b], "components": comps[b].reshape(-1)[: rc_ * d].reshape(rc_, d),
                 "evr": evr[b, :rc_], "k": int(k[b]), "Z": None} for b in range(B)]

    def acq_bounds(self):
        buf = self.acq_bounds_packed = np.zeros((self.B, 2 * self.max_d))     # (kept: sobol_draw_rows reads the boxes from it)
        self._chk(LIB.pcabo_batch_acq_bounds(self._h, _ptr(buf)))
        return [buf[b, : 2 * int(self.k[b])].reshape(2, int(self.k[b])).copy() for b in range(self.B)]

    def gp_wait_eval(self, Xq_list, best_f, maximize=False, acq=ACQ_LOG_EI):
        """Xq_list[b]: q x k_b points of run b.  Returns (values[B, q], status[B])."""
        q = Xq_list[0].shape[0]
        buf = np.zeros((self.B, q * self.max_d))
        for b, xq in enumerate(Xq_list):
            buf[b, : xq.size] = np.ascontiguousarray(xq, dtype=np.float64).ravel()
        bf = _f64(best_f, (self.B,))
        val = np.empty((self.B, q))
        status = np.zeros(self.B, dtype=np.int32)
        self._chk(LIB.pcabo_batch_gp_condition_end_eval(self._h, _ptr(buf), q, _ptr(bf), int(bool(maximize)), int(acq),
                                                        _ptr(val), _ptr(status)))
        return val, status

    # ---- the two waiting calls in halves (one thread advancing several batches: pcabo.batchrun.run_interleaved) -------------
    def busy(self) -> bool:
        rc = LIB.pcabo_batch_busy(self._h)
        if rc < 0:
            raise PcaboError(rc, self._err())
        return rc == 1

    def raw_row_buffer(self, q: int) -> np.ndarray:
        """The (B, q * max_d) buffer gp_eval_begin packs the runs' points into: a caller that writes run b's q x k_b points into
        `buf[b, :q*k_b].reshape(q, k_b)` itself (and passes those views) saves the copy."""
        buf = getattr(self, "_raw_buf", None)
        if buf is None or buf.shape != (self.B, q * self.max_d):
            buf = self._raw_buf = np.zeros((self.B, q * self.max_d))
        return buf

    def gp_eval_begin(self, Xq_list, best_f, maximize=False, acq=ACQ_LOG_EI):
        q = Xq_list[0].shape[0]
        buf = getattr(self, "_raw_buf", None)        # (only the first q * k_b entries of a row are read: no need to clear 5 MB per call)
        if buf is None or buf.shape != (self.B, q * self.max_d):
            buf = self._raw_buf = np.zeros((self.B, q * self.max_d))
        for b, xq in enumerate(Xq_list):
            if not (isinstance(xq, np.ndarray) and xq.base is buf):       # (raw_row_buffer: the caller drew straight into the row)
                buf[b, : xq.size] = np.ascontiguousarray(xq, dtype=np.float64).ravel()
        bf = _f64(best_f, (self.B,))
        self._chk(LIB.pcabo_batch_gp_condition_end_eval_begin(self._h, _ptr(buf), q, _ptr(bf), int(bool(maximize)), int(acq)))
        return (buf, q, bf, int(bool(maximize)), int(acq))

    def gp_eval_end(self, token):
        buf, q, bf, mx, acq = token
        val = np.empty((self.B, q))
        status = np.zeros(self.B, dtype=np.int32)
        self._chk(LIB.pcabo_batch_gp_condition_end_eval_end(self._h, _ptr(buf), q, _ptr(bf), mx, acq, _ptr(val), _ptr(status)))
        return val, status

    def _pack_ics(self, ics_list, bounds_list):
        B, MD = self.B, self.max_d
        nr = ics_list[0].shape[0]
        ics, bnd = np.zeros((B, nr * MD)), np.zeros((B, 2 * MD))
        for b in range(B):
            ics[b, : ics_list[b].size] = np.ascontiguousarray(ics_list[b], dtype=np.float64).ravel()
            bnd[b, : bounds_list[b].size] = np.ascontiguousarray(bounds_list[b], dtype=np.float64).ravel()
        return nr, ics, bnd

    def optimize_begin(self, ics_list, bounds_list, best_f, maximize=False, acq=ACQ_LOG_EI, batch_limit=5, maxiter=200):
        """Enqueue the device-resident optimisation; returns a token for optimize_end, or None when the call does not qualify
        (the caller then uses optimize_acqf)."""
        nr, ics, bnd = self._pack_ics(ics_list, bounds_list)
        bf = _f64(best_f, (self.B,))
        rc = LIB.pcabo_batch_optimize_acqf_begin(self._h, _ptr(ics), nr, int(batch_limit), _ptr(bnd), int(maxiter), _ptr(bf),
                                                 int(bool(maximize)), int(acq))
        if rc == 1:
            return None
        self._chk(rc)
        return (nr, int(batch_limit))

    def optimize_end(self, token):
        nr, batch_limit = token
        B, MD = self.B, self.max_d
        ng = (nr + batch_limit - 1) // batch_limit
        cand, vals = np.zeros((B, nr * MD)), np.zeros((B, nr))
        info = np.zeros((B, ng, 4), dtype=np.int32)
        failed, status = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._chk(LIB.pcabo_batch_optimize_acqf_end(self._h, nr, batch_limit, _ptr(cand), _ptr(vals), _ptr(info), _ptr(failed), _ptr(status)))
        out = []
        for b in range(B):
            k = int(self.k[b])
            out.append((cand[b, : nr * k].reshape(nr, k).copy(), vals[b].copy(), info[b].copy(), bool(failed[b])))
        return out, status

    def optimize_acqf(self, ics_list, bounds_list, best_f, maximize=False, acq=ACQ_LOG_EI, batch_limit=5, maxiter=200):
        """ics_list[b]: num_restarts x k_b; bounds_list[b]: 2 x k_b.  Returns per run (cand, vals, info, failed) + status."""
        B, MD = self.B, self.max_d
        nr = ics_list[0].shape[0]
        ng = (nr + batch_limit - 1) // batch_limit
        ics, bnd = np.zeros((B, nr * MD)), np.zeros((B, 2 * MD))
        for b in range(B):
            ics[b, : ics_list[b].size] = np.ascontiguousarray(ics_list[b], dtype=np.float64).ravel()
            bnd[b, : bounds_list[b].size] = np.ascontiguousarray(bounds_list[b], dtype=np.float64).ravel()
        bf = _f64(best_f, (B,))
        cand, vals = np.zeros((B, nr * MD)), np.zeros((B, nr))
        info = np.zeros((B, ng, 4), dtype=np.int32)
        failed, status = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._chk(LIB.pcabo_batch_optimize_acqf(self._h, _ptr(ics), nr, int(batch_limit), _ptr(bnd), int(maxiter), _ptr(bf),
                                                int(bool(maximize)), int(acq), _ptr(cand), _ptr(vals), _ptr(info),
                                                _ptr(failed), _ptr(status)))
        out = []
        for b in range(B):
            k = int(self.k[b])
            out.append((cand[b, : nr * k].reshape(nr, k).copy(), vals[b].copy(), info[b].copy(), bool(failed[b])))
        return out, status

    def device_acq_eval(self, xq_list, best_f, maximize=False, acq=ACQ_LOG_EI):
        """Value and gradient at xq_list[b] (q x k_b, q <= 32) through the evaluation of the device-resident optimiser."""
        B, MD = self.B, self.max_d
        q = xq_list[0].shape[0]
        xq = np.zeros((B, q * MD))
        for b in range(B):
            xq[b, : xq_list[b].size] = np.ascontiguousarray(xq_list[b], dtype=np.float64).ravel()
        bf = _f64(best_f, (B,))
        val, grad = np.zeros((B, q)), np.zeros((B, q * MD))
        self._chk(LIB.pcabo_batch_device_acq_eval(self._h, _ptr(xq), int(q), _ptr(bf), int(bool(maximize)), int(acq), _ptr(val), _ptr(grad)))
        return [val[b].copy() for b in range(B)], [grad[b, : q * int(self.k[b])].reshape(q, int(self.k[b])).copy() for b in range(B)]

    def set_profiling(self, on: bool) -> None:
        self._chk(LIB.pcabo_batch_set_profiling(self._h, int(bool(on))))

    def set_active(self, active) -> None:
        """active[b] = False parks run b: it stays in the lock-step launches but is skipped by optimize_acqf."""
        a = np.ascontiguousarray(active, dtype=np.int32).reshape(self.B)
        self._chk(LIB.pcabo_batch_set_active(self._h, _ptr(a)))

    def condition_profile(self) -> dict:
        """Device milliseconds of the last conditioning's phases (after it has been waited for)."""
        ms = np.zeros(4)
        self._chk(LIB.pcabo_batch_get_profile(self._h, _ptr(ms)))
        return dict(zip(("wpca", "gram", "cholesky", "root_inverse_alpha"), ms.tolist()))

    def inverse_map(self, z_list):
        z = np.zeros((self.B, self.max_d))
        for b, zb in enumerate(z_list):
            z[b, : zb.size] = np.asarray(zb, dtype=np.float64).ravel()
        x = np.empty((self.B, self.d))
        self._chk(LIB.pcabo_batch_inverse_map(self._h, _ptr(z), _ptr(x)))
        return x

    def inverse_map_begin(self, z_list) -> None:
        z = np.zeros((self.B, self.max_d))
        for b, zb in enumerate(z_list):
            z[b, : zb.size] = np.asarray(zb, dtype=np.float64).ravel()
        self._chk(LIB.pcabo_batch_inverse_map_begin(self._h, _ptr(z)))

    def inverse_map_end(self):
        x = np.empty((self.B, self.d))
        self._chk(LIB.pcabo_batch_inverse_map_end(self._h, _ptr(x)))
        return x


def sobol_scramble(state: np.ndarray, ltm: np.ndarray) -> None:
    """In-place scramble of a (k, 30) int64 Sobol state with (k, 30, 30) int64 lower-triangular bit matrices."""
    assert state.dtype == np.int64 and ltm.dtype == np.int64 and state.flags.c_contiguous and ltm.flags.c_contiguous
    rc = LIB.pcabo_sobol_scramble(_ptr(state), _ptr(ltm), int(state.shape[0]))
    if rc != 0:
        raise PcaboError(rc, "pcabo_sobol_scramble: bad argument")


def sobol_draw(state: np.ndarray, shift: np.ndarray, n: int, lo=None, rng=None, out=None) -> np.ndarray:
    """n points of a fresh scrambled engine ((k, 30) int64 state after `sobol_scramble`, (k,) int64 shift), optionally mapped
    into the box lo + rng * u - bit-identical to torch's SobolEngine.draw + botorch's scaling (pcabo_sobol_draw)."""
    assert state.dtype == np.int64 and shift.dtype == np.int64 and state.flags.c_contiguous and shift.flags.c_contiguous
    k = int(state.shape[0])
    if out is None:
        out = np.empty((int(n), k))
    else:
        assert out.shape == (int(n), k) and out.dtype == np.float64 and out.flags.c_contiguous
    if lo is not None:
        lo, rng = np.ascontiguousarray(lo, dtype=np.float64), np.ascontiguousarray(rng, dtype=np.float64)
    rc = LIB.pcabo_sobol_draw(_ptr(state), _ptr(shift), k, int(n), _ptr(lo), _ptr(rng), _ptr(out))
    if rc != 0:
        raise PcaboError(rc, "pcabo_sobol_draw: bad argument")
    return out


def sobol_draw_rows(engines, n: int, boxes: np.ndarray, outbuf: np.ndarray):
    """`sobol_draw` for the runs of a batch in one native call (pcabo_sobol_draw_rows): engines[b] (objects with .k, .state, .shift;
    None = skip) draws n points into `outbuf[b, :n*k_b]`, mapped into the box packed in `boxes[b]` as [lo(k_b), hi(k_b)]
    (Batch.acq_bounds_packed).  Returns the list of (n, k_b) views into `outbuf` (None for skipped runs)."""
    B = len(engines)
    assert boxes.dtype == np.float64 and boxes.flags.c_contiguous and boxes.shape[0] == B
    assert outbuf.dtype == np.float64 and outbuf.flags.c_contiguous and outbuf.shape[0] == B
    ks = np.zeros(B, dtype=np.int32)
    st, sh, ou = (C.c_void_p * B)(), (C.c_void_p * B)(), (C.c_void_p * B)()
    base, stride = outbuf.ctypes.data, outbuf.strides[0]
    views = [None] * B
    for b, e in enumerate(engines):
        if e is None:
            continue
        k = ks[b] = e.k
        assert 2 * k <= boxes.shape[1] and n * k <= outbuf.shape[1] and e.state.dtype == np.int64 and e.shift.dtype == np.int64
        st[b], sh[b], ou[b] = e.state.ctypes.data, e.shift.ctypes.data, base + b * stride
        views[b] = outbuf[b, : n * k].reshape(n, k)
    rc = LIB.pcabo_sobol_draw_rows(st, sh, _ptr(ks), B, int(n), _ptr(boxes), boxes.strides[0] // 8, ou)
    if rc != 0:
        raise PcaboError(rc, "pcabo_sobol_draw_rows: bad argument")
    return views


def comm_unique_id() -> bytes:
    """128 bytes (ncclUniqueId) from rank 0 for `Comm`; the caller distributes them to the other ranks."""
    buf = C.create_string_buffer(128)
    rc = LIB.pcabo_comm_unique_id(buf)
    if rc != 0:
        raise PcaboError(rc, "librccl.so could not be opened or ncclGetUniqueId failed")
    return buf.raw


class Comm:
    """RCCL communicator for the final gather of best-so-far values (pcabo_comm_* / pcabo_gather_best of include/pcabo.h)."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int = 0):
        self._h = C.c_void_p()
        rc = LIB.pcabo_comm_create(C.c_char_p(unique_id), int(world), int(rank), int(device), C.byref(self._h))
        if rc != 0:
            msg = self._err() if self._h else "librccl.so could not be opened"
            if self._h:
                LIB.pcabo_comm_destroy(self._h)
                self._h = C.c_void_p()
            raise PcaboError(rc, msg)
        self.world, self.rank = int(world), int(rank)

    def _err(self) -> str:
        buf = C.create_string_buffer(256)
        LIB.pcabo_comm_last_error(self._h, buf, 256)
        return buf.value.decode(errors="replace")

    def gather_best(self, local) -> np.ndarray:
        local = _f64(local).reshape(-1)
        out = np.empty((self.world, local.shape[0]))
        rc = LIB.pcabo_gather_best(self._h, _ptr(local), int(local.shape[0]), _ptr(out))
        if rc != 0:
            raise PcaboError(rc, self._err())
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            LIB.pcabo_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def lbfgsb_set_vector_kernels(enabled: bool) -> bool:
    """Host L-BFGS-B: AVX2 (default) or scalar O(m n) loops - same iterates; returns the previous setting (tests)."""
    return bool(LIB.pcabo_lbfgsb_set_vector_kernels(int(bool(enabled))))


def lbfgsb_set_sum_order(order: int) -> int:
    """0: the published summation order (scipy's iterates), 1: the device optimiser's 64-lane tree order.  Returns the previous one."""
    return int(LIB.pcabo_lbfgsb_set_sum_order(int(order)))


def lbfgsb_minimize(fun, x0, bounds, m=10, factr=1e7, pgtol=1e-5, maxiter=15000, maxfun=15000, maxls=20):
    """Host-only entry: this library's L-BFGS-B on a Python objective `fun(x) -> (f, g)` (for tests)."""
    x = _f64(x0).reshape(-1).copy()
    nvar = x.shape[0]
    lo = _f64([b[0] if b[0] is not None else -np.inf for b in bounds])
    hi = _f64([b[1] if b[1] is not None else np.inf for b in bounds])

    def cb(xp, gp, _user):
        xv = np.ctypeslib.as_array(xp, shape=(nvar,))
        f, g = fun(xv.copy())
        np.ctypeslib.as_array(gp, shape=(nvar,))[:] = g
        return float(f)

    cfun = FG_CALLBACK(cb)
    f_out, nit, nfev, task = C.c_double(0), C.c_int(0), C.c_int(0), C.c_int(0)
    warn = LIB.pcabo_lbfgsb_minimize(nvar, _ptr(x), _ptr(lo), _ptr(hi), cfun, None, int(m), float(factr), float(pgtol),
                                     int(maxiter), int(maxfun), int(maxls), C.byref(f_out), C.byref(nit),
                                     C.byref(nfev), C.byref(task))
    return {"x": x, "fun": f_out.value, "nit": nit.value, "nfev": nfev.value, "warnflag": warn, "task": task.value}
