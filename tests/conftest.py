import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_rowmax_err(row_ptr, got, ref):
    """SURVEY.md gotcha 14: per-row max|dK| / max|K_row| (entry-wise relative error is undefined on
    exact-cancellation entries)."""
    import numpy as np
    nrow = len(row_ptr) - 1
    lens = np.diff(row_ptr)
    rows = np.repeat(np.arange(nrow), lens)
    num = np.zeros(nrow); den = np.zeros(nrow)
    np.maximum.at(num, rows, np.abs(got - ref))
    np.maximum.at(den, rows, np.abs(ref))
    ok = den > 0
    worst = (num[ok] / den[ok]).max() if ok.any() else 0.0
    if (~ok).any():
        assert num[~ok].max() == 0.0 if (~ok & (lens > 0)).any() else True
    return worst
