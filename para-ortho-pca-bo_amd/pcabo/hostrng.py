"""torch's CPU generator for the host's pacing thread, without torch calls in the loop.

The reference's restart heuristic (botorch's gen_batch_initial_conditions behind PCA_BO.py:607-614) draws from torch's global
CPU generator: the Sobol scramble bits (`torch.randint(2, ...)`, twice per BO iteration) and the Boltzmann pick
(`torch.multinomial(weights, 10)`).  A lock-step driver keeps one generator per run; with a few hundred runs on one host
thread those calls were 40 % of the thread's time.  `HostMT` holds a run's generator as the state blob torch itself exports
(`torch.Generator.get_state()`, 5056 bytes: a 32-bit Mersenne Twister) and advances it with libpcabo's host helpers
(`pcabo_torch_randint2`, `pcabo_torch_multinomial_rows`, csrc/host_entry.cpp) - same numbers, same consumption, so the blob can
go back into a real `torch.Generator` at any time (`torch_generator()`): the rare paths (randperm when all values tie, botorch's
non-negative variant) still run torch's own code.

`NATIVE_OK` is established on first use by comparing both helpers with torch itself (picks, bits and the state blob after the
call); if anything differs - another torch build, another exponential sampler - every HostMT falls back to torch's calls on a
real generator.  tests/test_abi_and_host.py repeats the comparison on 10 000 weight vectors.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native

BLOB_BYTES = 5056
_LIB = _native.LIB
_LIB.pcabo_torch_randint2.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
_LIB.pcabo_torch_multinomial_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
_LIB.pcabo_boltzmann_pick_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]

_native_ok: Optional[bool] = None


def _raw_randint2(blob: np.ndarray, count: int) -> np.ndarray:
    out = np.empty(count, dtype=np.int64)
    rc = _LIB.pcabo_torch_randint2(blob.ctypes.data, int(count), out.ctypes.data)
    if rc != 0:
        raise _native.PcaboError(rc, "pcabo_torch_randint2: not a state blob of torch's CPU generator")
    return out


def _raw_multinomial_rows(blobs: Sequence[Optional[np.ndarray]], weights: np.ndarray, n_pick: int) -> np.ndarray:
    rows, n = weights.shape
    ptrs = (C.c_void_p * rows)(*[(b.ctypes.data if b is not None else None) for b in blobs])
    out = np.zeros((rows, n_pick), dtype=np.int64)
    rc = _LIB.pcabo_torch_multinomial_rows(ptrs, weights.ctypes.data, rows, n, int(n_pick), out.ctypes.data)
    if rc != 0:
        raise _native.PcaboError(rc, "pcabo_torch_multinomial_rows: bad arguments")
    return out


def native_ok() -> bool:
    """The helpers reproduce THIS torch build (checked once: bits of randint, picks of multinomial, the generator's state after)."""
    global _native_ok
    if _native_ok is None:
        try:
            ok = True
            for seed in (0, 43, 15407):
                g = torch.Generator().manual_seed(seed)
                blob = g.get_state().numpy().copy()
                ok &= blob.nbytes == BLOB_BYTES
                want = torch.randint(2, (700,), generator=g).numpy()           # (crosses a refill of the twister's 624 words)
                got = _raw_randint2(blob, 700)
                ok &= bool(np.array_equal(want, got))
                w = torch.rand(512, dtype=torch.float64, generator=torch.Generator().manual_seed(seed + 1)).exp()
                want_idx = torch.multinomial(w, 10, generator=g).numpy()
                got_idx = _raw_multinomial_rows([blob], np.ascontiguousarray(w.numpy()[None]), 10)[0]
                ok &= bool(np.array_equal(want_idx, got_idx)) and bool(np.array_equal(g.get_state().numpy(), blob))
            _native_ok = bool(ok)
        except Exception:                  # noqa: BLE001 - any surprise means: use torch
            _native_ok = False
    return _native_ok


class HostMT:
    """One run's torch CPU generator as a state blob (see the module docstring).  Quacks like the part of `torch.Generator`
    the lock-step drivers use: `get_state()` / `set_state()`."""
    __slots__ = ("blob", "_gen")

    def __init__(self, seed: int):
        g = torch.Generator().manual_seed(int(seed))
        if native_ok():
            self.blob, self._gen = g.get_state().numpy().copy(), None
        else:                               # fallback: a real generator, torch's own calls
            self.blob, self._gen = None, g

    # ---- the torch.Generator surface the drivers use ---------------------------------------------------------------
    def get_state(self) -> torch.Tensor:
        return self._gen.get_state() if self._gen is not None else torch.from_numpy(self.blob.copy())

    def set_state(self, state) -> None:
        if self._gen is not None:
            self._gen.set_state(state)
        else:
            self.blob[:] = state.numpy() if isinstance(state, torch.Tensor) else np.asarray(state, dtype=np.uint8)

    # ---- draws ---------------------------------------------------------------------------------------------------------
    def randint2(self, shape) -> np.ndarray:
        """`torch.randint(2, shape, generator=g).numpy()`"""
        if self._gen is not None:
            return torch.randint(2, tuple(shape), generator=self._gen).numpy()
        return _raw_randint2(self.blob, int(np.prod(shape))).reshape(shape)

    def torch_generator(self) -> torch.Generator:
        """A real generator in this state, for the rare torch-only paths; hand it back with `absorb`."""
        if self._gen is not None:
            return self._gen
        g = torch.Generator()
        g.set_state(torch.from_numpy(self.blob))
        return g

    def absorb(self, g: torch.Generator) -> None:
        if self._gen is None:
            self.blob[:] = g.get_state().numpy()


def multinomial_rows(weights: np.ndarray, n_pick: int, generators: Sequence, rows: Sequence[int]) -> dict:
    """`torch.multinomial(weights[b], n_pick, generator=generators[b]).numpy()` for b in rows -> {b: indices}.  HostMT
    generators go through one native call, real torch generators through torch."""
    out = {}
    nat = [b for b in rows if isinstance(generators[b], HostMT) and generators[b]._gen is None]
    if nat:
        w = np.ascontiguousarray(weights[nat], dtype=np.float64)
        idx = _raw_multinomial_rows([generators[b].blob for b in nat], w, n_pick)
        for j, b in enumerate(nat):
            out[b] = idx[j]
    for b in rows:
        if b not in out:
            g = generators[b]
            g = g._gen if isinstance(g, HostMT) else g
            out[b] = torch.multinomial(torch.from_numpy(np.ascontiguousarray(weights[b])), n_pick, generator=g).numpy()
    return out


def boltzmann_pick_rows(vals: np.ndarray, n_pick: int, eta: float, generators: Sequence, skip=()):
    """botorch's initialize_q_batch for every row of `vals` in ONE native call (pcabo_boltzmann_pick_rows) when every generator is
    a state blob; returns (indices [B x n_pick], flags [B]: 0 picked, 1 all values equal - nothing drawn, 2 skipped) or None when a
    generator is a real torch generator (the caller keeps its torch path)."""
    if not native_ok() or any(not isinstance(g, HostMT) or g._gen is not None for g in generators):
        return None
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    rows, n = vals.shape
    skip = set(skip)
    ptrs = (C.c_void_p * rows)(*[(None if b in skip else generators[b].blob.ctypes.data) for b in range(rows)])
    out = np.zeros((rows, n_pick), dtype=np.int64)
    flags = np.zeros(rows, dtype=np.int32)
    rc = _LIB.pcabo_boltzmann_pick_rows(ptrs, vals.ctypes.data, rows, n, int(n_pick), float(eta), out.ctypes.data, flags.ctypes.data)
    if rc != 0:
        raise _native.PcaboError(rc, "pcabo_boltzmann_pick_rows: bad arguments")
    return out, flags
