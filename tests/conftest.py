import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


def _gpu_present() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently; `-m "not gpu"` never imports HIP.
    pass


@pytest.fixture(scope='session')
def golden_dir() -> Path:
    return ROOT / 'tests' / 'golden'
