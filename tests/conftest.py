import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/`."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _mi_env_follows_monkeypatch(monkeypatch):
    """The library and the package read their MI_* A/B switches once (mi_env_reload / image_restoration_amd.reload_env re-read
    them): every monkeypatch.setenv / delenv inside a test, and the restore at its end, is followed by a reload, so the tests
    keep flipping switches per call as before."""
    def reload():
        mod = sys.modules.get("image_restoration_amd")
        if mod is not None and hasattr(mod, "reload_env"):
            mod.reload_env()

    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        orig_set(name, value, prepend)
        reload()

    def delenv(name, raising=True):
        orig_del(name, raising)
        reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()          # restore the environment now (idempotent: monkeypatch's own teardown finds nothing left) ...
    reload()                    # ... and let the library see it
