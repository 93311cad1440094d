import os
import sys

# PyTorch-ROCm bundles its own HIP runtime: in a process that uses both, torch has to be loaded before the engine's
# library (whose dependency on libamdhip64 is then satisfied by the copy torch brought), or torch finds no device.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
