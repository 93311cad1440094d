"""MI355X engine for the BPR-SGD step and the full-catalog evaluation of sequential recommenders.

The product is the HIP library recommendation_amd/librecsys_amd.so behind the C ABI declared in
include/recsys_engine.h; this package is the thin ctypes view of it used by the tests and by
bench.py.  There is no CPU fallback: importing `capi` without the built library raises."""
from .capi import Engine, Metrics, MODEL_ID, build_rebus_paths, fsub_mine, lib  # noqa: F401
