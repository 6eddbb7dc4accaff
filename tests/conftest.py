import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    if not hasattr(config, "workerinput"):
        # the pytest-xdist controller (pytest.ini: -n 4), or a single-process run: build once BEFORE the workers start (they then find every library fresh instead of racing
        # for the same object files) and map libhipadj.so into this process too — the tests run in the workers, but a record of "which native libraries did the pytest
        # processes load" that looks at the controller must see the library the suite is about
        try:
            import scimlsensitivity_jl_amd as mod
            mod.build_extension()
            mod.load_library()
        except Exception as e:      # noqa: BLE001 — the `sa` fixture reports a broken build where it matters
            sys.stderr.write(f"conftest: library not built / loaded in the controller: {e!r}\n")


def _hip_device_count():
    """Is a ROCm GPU visible?  Read from the kernel driver's device node — not through a HIP runtime: loading /opt/rocm's
    libamdhip64 into the test process next to the one torch bundles would give the process two HIP runtimes."""
    return 1 if os.path.exists("/dev/kfd") else 0


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without a GPU: the `gpu`-marked tests are skipped at collection time (the product has no CPU
    fallback, they could only fail with HIPADJ_ERR_NO_DEVICE).  `-m gpu` on a box that HAS a device is unaffected."""
    if not any("gpu" in it.keywords for it in items) or _hip_device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible (there is no CPU fallback)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "gradients.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def sa():
    import scimlsensitivity_jl_amd as mod
    mod.build_extension()
    return mod
