"""pytest plugin: the whole GPU suite with LDS + register files refilled with a pattern before every worker library call.

    PYTHONPATH=tests POISON_PAT=0xffffffff python -m pytest -p tools.poison_plugin tests -m gpu -q --deselect tests/test_gpu_multirank.py --deselect tests/test_gpu_uninit.py

(POISON_WHAT: bit 0 LDS, bit 1 registers; default 3.  POISON_LEVEL=kernels: in front of every kernel launch inside the library instead of
every library call.  The multi-rank tests start their own processes, which the plugin does not reach.)"""
import importlib
import os


def pytest_configure(config):
    from __graft_entry__ import load_package
    from tools import poison
    pkg = load_package()
    poison.build()
    binding = importlib.import_module(pkg.__name__ + ".binding")
    engine = importlib.import_module(pkg.__name__ + ".host.engine")
    pat = int(os.environ.get("POISON_PAT", "0xffffffff"), 0)
    what = int(os.environ.get("POISON_WHAT", "3"))
    if os.environ.get("POISON_LEVEL", "calls") == "kernels":
        config._poison_cm = poison.poisoned_kernel_launches(binding, pat, what)
    else:
        config._poison_cm = poison.poisoned_worker_calls(binding, engine, pat, what)
    config._poison_calls = config._poison_cm.__enter__()
    print("poison plugin: pattern %#x what %d" % (pat, what), flush=True)


def pytest_unconfigure(config):
    cm = getattr(config, "_poison_cm", None)
    if cm is not None:
        print("poison plugin: %d poisoned calls / launches" % config._poison_calls[0], flush=True)
        cm.__exit__(None, None, None)
