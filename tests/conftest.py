import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the fault injection of the persistent launches (TP_OPT_INJECT_GIVE_UP) is a test instrument: the library refuses it without this
os.environ.setdefault("TPOSE_ALLOW_FAULT_INJECTION", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
