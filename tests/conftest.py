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


@pytest.fixture(autouse=True)
def _gpu_tests_fail_loudly_without_a_gpu(request):
    """`-m gpu` on a box without a GPU must FAIL, never skip or pass on some fallback: every gpu-marked test first asks the
    product for its device (engine.require_gpu raises YkError when the library or the device is missing)."""
    if request.node.get_closest_marker('gpu') is not None:
        from k210_yolo_framework_amd import engine
        engine.require_gpu()
    yield


@pytest.fixture(scope='session')
def golden_dir() -> Path:
    return ROOT / 'tests' / 'golden'
