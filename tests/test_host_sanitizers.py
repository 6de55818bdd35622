"""CPU-side guards (SURVEY.md section 5 "race detection / sanitizers").

1. The host side of the library - csrc/lbfgsb.cpp, csrc/host_entry.cpp, csrc/host_side.h (RestartGroup, GangPool) and
   csrc/lb_plan.h - built with g++ under AddressSanitizer, UndefinedBehaviorSanitizer and ThreadSanitizer (`make asan ubsan tsan`,
   no GPU, no HIP runtime) and run: a driver with a stub launcher, and the scipy comparison of tests/test_lbfgsb_vs_scipy.py once
   more through the ASan build of the entry points.
2. The kernarg segment of every kernel in the built code objects stays below 3 968 bytes: round 2's abort under the tracer came
   with a segment of exactly HIP's 4 096-byte maximum (EXPERIMENTS.md section 5), and nothing guarded that bound.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "para-ortho-pca-bo_amd", "csrc")
LIBDIR = os.path.join(ROOT, "para-ortho-pca-bo_amd", "lib")


@pytest.mark.parametrize("san", ["asan", "ubsan", "tsan"])
def test_host_side_under_sanitizer(san):
    env = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run(["make", "-C", CSRC, san], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:]
    assert "host selftest ok" in out.stdout, out.stdout[-2000:]
    assert "Sanitizer" not in out.stdout, out.stdout[-4000:]


_CHILD = r"""
import ctypes as C, sys
import numpy as np
from scipy.optimize import minimize
lib = C.CDLL(sys.argv[1])
FG = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
lib.pcabo_lbfgsb_minimize.restype = C.c_int
lib.pcabo_lbfgsb_minimize.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), FG, C.c_void_p,
    C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
def ptr(a): return a.ctypes.data_as(C.POINTER(C.c_double))
def mine(fun, x0, lo, hi, maxiter):
    x = np.array(x0, dtype=np.float64); n = x.size
    def cb(xp, gp, _u):
        f, g = fun(np.ctypeslib.as_array(xp, shape=(n,)).copy())
        np.ctypeslib.as_array(gp, shape=(n,))[:] = g
        return float(f)
    f, nit, nfev, task = C.c_double(0), C.c_int(0), C.c_int(0), C.c_int(0)
    w = lib.pcabo_lbfgsb_minimize(n, ptr(x), ptr(lo), ptr(hi), FG(cb), None, 10, 1e7, 1e-5, maxiter, 15000, 20, C.byref(f), C.byref(nit), C.byref(nfev), C.byref(task))
    return x, nit.value, nfev.value, w
def rosen(x):
    f = np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g
rng = np.random.default_rng(1)
same = 0
for _ in range(25):
    n = int(rng.integers(2, 30))
    x0 = rng.uniform(-2, 2, n); lo = rng.uniform(-3, 0.5, n); hi = lo + rng.uniform(0.5, 4, n)
    mi = int(rng.integers(5, 300))
    ref = minimize(rosen, x0, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)), options={"maxiter": mi})
    x, nit, nfev, w = mine(rosen, x0, lo, hi, mi)
    assert (ref.nit, ref.nfev) == (nit, nfev), (ref.nit, ref.nfev, nit, nfev)
    assert np.abs(ref.x - x).max() < 1e-8
    same += 1
print("scipy comparison under the sanitizer ok:", same)
"""


def test_scipy_comparison_through_the_asan_build():
    """The 25 bounded Rosenbrock problems of test_lbfgsb_vs_scipy.py through libpcabo_host_asan.so (the interpreter runs with the
    sanitizer runtime preloaded): same iteration / evaluation counts as scipy, no report."""
    out = subprocess.run(["make", "-C", CSRC, os.path.join("..", "lib", "libpcabo_host_asan.so")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:]
    rt = subprocess.run(["g++", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no libasan.so next to this g++")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="halt_on_error=1:detect_leaks=0", OMP_NUM_THREADS="1")
    run = subprocess.run([sys.executable, "-c", _CHILD, os.path.join(LIBDIR, "libpcabo_host_asan.so")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-4000:]
    assert "scipy comparison under the sanitizer ok: 25" in run.stdout, run.stdout[-2000:]
    assert "AddressSanitizer" not in run.stdout, run.stdout[-4000:]


KERNARG_LIMIT = 3968      # bytes; HIP's maximum is 4 096


def _kernarg_sizes(path):
    """(kernel name, kernarg_segment_size) of every kernel in the gfx950 code objects bundled in `path` (ELF notes)."""
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    unbundle = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    import tempfile
    sizes = []
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        # the device code of a translation unit is a bundle inside the object's .hip_fatbin section
        r = subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return None
        r = subprocess.run([unbundle, "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                            "--output=" + co], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return None
        notes = subprocess.run([readelf, "--notes", co], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    name = None
    for line in notes.splitlines():
        m = re.search(r"\.kernarg_segment_size:\s*(\d+)", line)
        if m:
            pending = int(m.group(1))
            sizes.append([None, pending])
        m = re.search(r"^\s*\.name:\s*(\S+)", line)
        if m and sizes and sizes[-1][0] is None:
            sizes[-1][0] = m.group(1)
    return [(n or "?", s) for n, s in sizes]


def test_kernarg_segments_stay_below_the_limit(native):
    objs = [os.path.join(LIBDIR, f) for f in sorted(os.listdir(LIBDIR)) if f.startswith("kernels_") and f.endswith(".o")]
    objs.append(os.path.join(LIBDIR, "pcabo_api.o"))
    assert len(objs) >= 6
    seen = 0
    worst = (None, 0)
    for o in objs:
        sizes = _kernarg_sizes(o)
        if sizes is None:
            continue                                # (an object without device code)
        for name, size in sizes:
            seen += 1
            if size > worst[1]:
                worst = (name, size)
            assert size <= KERNARG_LIMIT, "%s: kernarg segment of %d bytes (limit %d, HIP's maximum 4096) in %s" % (name, size, KERNARG_LIMIT, o)
    assert seen >= 27, "only %d kernels found in the code objects" % seen
    print("kernels: %d, largest kernarg segment: %s = %d bytes" % (seen, worst[0], worst[1]))
