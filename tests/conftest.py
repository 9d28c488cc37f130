import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a hung kernel or rendezvous must fail the test, not eat the whole GPU slot (pytest-timeout, when installed)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _box_identity(request):
    """GPU sessions print where they ran (a failure that follows one box around is a hardware / driver question, not a code one)."""
    if "gpu" not in (request.config.getoption("-m") or "") or "not gpu" in (request.config.getoption("-m") or ""):
        yield
        return
    import socket
    import subprocess
    try:
        uid = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=20).stdout
        uid = " ".join(ln.split(":")[-1].strip() for ln in uid.splitlines() if "Unique ID" in ln and "GPU[" in ln)
    except Exception as e:  # noqa: BLE001
        uid = f"rocm-smi failed: {e}"
    cpu = next((ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")
    print(f"\n[box] host {socket.gethostname()} | {os.cpu_count()} x {cpu} | GPU unique id {uid}", flush=True)
    yield
