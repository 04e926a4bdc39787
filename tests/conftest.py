import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The product library bound to GPU 0.  Fails loudly (no fallback) when unavailable."""
    from madsim_amd import runtime
    # A box with no AMD GPU device node at all (this build container) cannot run the gpu tests: skip them there so a
    # plain `pytest tests` is green; on a GPU box (or with MADSIM_REQUIRE_GPU=1) a missing device is a loud failure.
    if not os.path.exists("/dev/kfd") and os.environ.get("MADSIM_REQUIRE_GPU", "0") != "1":
        pytest.skip("no GPU device node (/dev/kfd): gpu tests need a real MI355X")
    # Some gpu tests hand torch buffers / streams to the library.  torch's wheel bundles its own ROCm runtime; a process that
    # loads /opt/rocm's first (through libmadsim_hip.so) and torch's second ends up with two, and the second finds "No HIP GPUs".
    # Importing torch first makes both resolve to one copy — the order bench.py has anyway.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    runtime.init(0)
    yield runtime
    runtime.shutdown()


def pytest_terminal_summary(terminalreporter):
    """The fuzz suites' account of themselves: per generator, seeds compared / re-run with grown capacities after a first-pass
    MADSIM_OVERFLOW / proven beyond the layout's ceilings (tests/parity.py: the only seeds that are not compared)."""
    for modname in ("tests.test_gpu_parity", "tests.test_emu_parity"):
        mod = sys.modules.get(modname)
        t = getattr(mod, "TALLY", None)
        if t is not None and t.rows:
            terminalreporter.write_line(f"[{modname}] fuzz parity: {t.n} seeds compared with the oracle, {t.rerun} of them after a re-run "
                                        f"with grown capacities, {t.unresolved} proven beyond the layout's ceilings")
            for k, v in sorted(t.rows.items()):
                terminalreporter.write_line(f"    {k}: seeds {v[0]}, re-run {v[1]}, beyond ceilings {v[2]}")
