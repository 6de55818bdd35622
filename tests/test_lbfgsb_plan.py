"""The work plan of the device-resident optimiser's triangular passes (csrc/kernels_lbfgsb.hip: lb_build_plan), run on the host
through the library's debug entry: for every n the segments of a pass cover every unit's range exactly once, a wave has at most
two segments, the partial slots of a unit are consecutive and ascending, and their number stays within what the launch reserves
in LDS (which stays below the 150 KB the kernel's attribute allows).  No GPU needed."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def plan_fn(native):
    fn = native.LIB.pcabo_debug_lbfgsb_plan
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    return fn


def test_plan_covers_every_unit_once_with_bounded_slots(plan_fn):
    out = np.zeros(4096, dtype=np.int32)
    sizes = np.zeros(4, dtype=np.int32)
    worst = 0
    for n in range(1, 513):
        NP = 64 * ((n + 63) // 64)
        S = NP // 64
        assert plan_fn(n, NP, out.ctypes.data, sizes.ctypes.data) == 0
        per_pass, per_wave, max_slots, lds = (int(v) for v in sizes)
        assert lds <= 150 * 1024
        for pas in range(2):
            base = pas * per_pass
            lo = [0 if pas == 0 else 64 * u for u in range(S)]
            hi = [min(n, 64 * (u + 1)) if pas == 0 else max(n, 64 * u) for u in range(S)]
            cover = {u: [] for u in range(S)}
            slots = 0
            for w in range(16):
                e = out[base + w * per_wave: base + (w + 1) * per_wave]
                assert 0 <= e[0] <= 2
                work = 0
                for g in range(e[0]):
                    u, a, b, dest = (int(v) for v in e[1 + 4 * g: 5 + 4 * g])
                    assert 0 <= u < S and lo[u] <= a < b <= hi[u] and dest >= -1
                    cover[u].append((a, b, dest))
                    slots += dest >= 0
                    work += b - a
                worst = max(worst, work)
            assert slots <= max_slots, (n, pas, slots, max_slots)
            table = out[base + 16 * per_wave: base + 16 * per_wave + 2 * S]
            for u in range(S):
                c = sorted(cover[u])
                assert c and c[0][0] == lo[u] and c[-1][1] == hi[u] and c[0][2] == -1, (n, pas, u, c)
                for i in range(1, len(c)):
                    assert c[i][0] == c[i - 1][1]
                extra = [x[2] for x in c[1:]]
                assert int(table[2 * u + 1]) == len(extra)
                if extra:
                    assert extra == list(range(int(table[2 * u]), int(table[2 * u]) + len(extra))), (n, pas, u, extra)
    assert worst <= 144          # (a slab per wave pair left the longest wave with 225 columns at n = 449)


def test_plan_rejects_sizes_the_kernel_does_not_take(plan_fn):
    out = np.zeros(4096, dtype=np.int32)
    sizes = np.zeros(4, dtype=np.int32)
    assert plan_fn(10, 128, out.ctypes.data, sizes.ctypes.data) == -1        # NP is not n rounded up to 64
    assert plan_fn(600, 640, out.ctypes.data, sizes.ctypes.data) == -1       # beyond the kernel's 512
