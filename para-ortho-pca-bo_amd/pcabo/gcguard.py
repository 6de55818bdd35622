"""Keep CPython's cyclic garbage collector out of the BO loop.

With torch and numpy imported a process holds several hundred thousand long-lived container objects.  The loop of a run
allocates a few hundred tracked objects per iteration (tensors, arrays, tuples), which trips the collector's generation
thresholds regularly, and every full collection walks all of those old objects again: measured on the MI355X host as
0.3-0.45 ms per BO iteration (of 3.0-3.6), landing wherever the threshold happens to trip - typically in the torch ops of
the initial pick.  `gc.freeze()` moves everything that exists when a run starts into the permanent generation, so the
collections during the run only see the run's own objects; `gc.unfreeze()` gives the objects back when the last run
ends.  Nothing is leaked and the collector stays enabled.  The optimisers' `gc_freeze=False` switches this off.
"""
from __future__ import annotations

import gc
import threading

_lock = threading.Lock()
_depth = 0
_saved_threshold = None
_YOUNG_THRESHOLD = 50000


def enter() -> None:
    global _depth, _saved_threshold
    with _lock:
        if _depth == 0:
            gc.freeze()
            _saved_threshold = gc.get_threshold()
            gc.set_threshold(max(_saved_threshold[0], _YOUNG_THRESHOLD), *_saved_threshold[1:])
        _depth += 1


def leave() -> None:
    global _depth, _saved_threshold
    with _lock:
        if _depth > 0:
            _depth -= 1
            if _depth == 0:
                if _saved_threshold is not None:
                    gc.set_threshold(*_saved_threshold)
                    _saved_threshold = None
                gc.unfreeze()
