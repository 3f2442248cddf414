import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(autouse=True)
def _restore_context_options(request):
    """options a test set on a (module-scoped) Device go back to what they were, also when the test failed"""
    yield
    for name in ("dev", "dev_shfl"):
        d = request.node.funcargs.get(name) if hasattr(request.node, "funcargs") else None
        if d is not None and hasattr(d, "restore_options"):
            d.restore_options()


def pytest_terminal_summary(terminalreporter):
    """Cross-checks against development-variant kernels that this build of libvszip_hip.so does not contain are not silent
    (ADVICE r5): the suite says how many it left out, by switch. A -DVSZIP_DEV_VARIANTS build runs them all."""
    try:
        import vszip_amd

        absent = dict(vszip_amd.capi.VARIANTS_ABSENT)
    except Exception:
        return
    if absent:
        terminalreporter.write_line(
            "vszip: %d variant cross-checks not run (switches this build does not contain: %s)"
            % (sum(absent.values()), ", ".join(f"{k} x{v}" for k, v in sorted(absent.items()))))
