"""CPU oracle for the Restormer / MoCE-IR transformer-block hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``image_restoration_amd/`` may import
this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and there only as the checker.

The oracle is a plain-PyTorch (CPU, fp32/fp64) functional restatement of the
reference's algorithm; every function cites the reference file:line it follows
(paths relative to the upstream repo root).  It is pinned against golden
vectors captured from the imported reference itself (``tools/capture_golden.py``
-> ``tests/golden/*.npz``, checked by ``tests/test_oracle_golden.py``).
"""
from .restormer_ref import *  # noqa: F401,F403
