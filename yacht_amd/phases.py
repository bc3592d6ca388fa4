"""Wall-clock phase accounting of the `yacht train` / `yacht run` commands (bench_e2e.py reads it).

    with phases.phase("unzip"): ...

Nested phases are kept separately ("train_core/read_sig_files" is also inside "train_core"); the
accounting costs two perf_counter calls per phase and is always on."""
from __future__ import annotations

import time
from contextlib import contextmanager
from typing import Dict

TIMES: Dict[str, float] = {}
_stack = []


def reset() -> None:
    TIMES.clear()
    _stack.clear()


@contextmanager
def phase(name: str):
    full = "/".join(_stack + [name])
    _stack.append(name)
    t0 = time.perf_counter()
    try:
        yield
    finally:
        _stack.pop()
        TIMES[full] = TIMES.get(full, 0.0) + time.perf_counter() - t0


def snapshot() -> Dict[str, float]:
    return {k: round(v, 4) for k, v in TIMES.items()}
