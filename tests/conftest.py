import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the parity tests inspect stages on quiet handles too (include/shk.h: shk_new, `verbose`): keep the stage data everywhere
    os.environ.setdefault("SHK_KEEP_STAGES", "1")
    # ... and several tests tell from the stage timers which path a handle took ("count_dedupe_kernel", "batch_pack_kernel",
    # "shard_graph_stitch", "device_writer_kernels"): quiet handles record them too here (bench.py runs without them)
    os.environ.setdefault("SHK_STAGE_TIMERS", "1")
    # torch bundles its own libamdhip64 (same soname): import it BEFORE libshk_hip.so is loaded so
    # that one HIP runtime serves both and device pointers can be shared (bench.py does the same)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    # build what is missing (the driver normally ran __graft_entry__.build() already)
    so = os.path.join(ROOT, "sparrowhawk_amd", "libshk_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sparrowhawk_amd", "csrc")],
                              stdout=subprocess.DEVNULL)
    from oracle import build_oracle
    build_oracle()


@pytest.fixture(scope="session")
def lib():
    from sparrowhawk_amd import _lib
    return _lib.load()
