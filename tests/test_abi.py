"""CPU-side checks of the drop-in boundary: libdpmmhip.so loads, exports every symbol that
include/dpmm_hip.h (+ dpmm_hip_master.h, dpmm_hip_debug.h) declare, and refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from __graft_entry__ import ROOT, load_package


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    p.build_library()
    return p


WORKER_HEADERS = ("dpmm_hip.h", "dpmm_hip_master.h", "dpmm_hip_debug.h")     # drop-in surface | optional device master | diagnostics


def declared_in(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dpmm_[a-z0-9_]+)\s*\(", src)))


def declared_functions():
    return sorted(set(n for h in WORKER_HEADERS for n in declared_in(h)))


def test_the_three_worker_headers_partition_the_abi(pkg):
    """Every exported entry point is declared in exactly one header; the drop-in header carries no device-master and no diagnostic entry."""
    per = {h: declared_in(h) for h in WORKER_HEADERS}
    allnames = [n for h in WORKER_HEADERS for n in per[h]]
    assert len(allnames) == len(set(allnames))
    assert not [n for n in per["dpmm_hip.h"] if "_master" in n or n.startswith("dpmm_debug_") or n.startswith("dpmm_last_") and n != "dpmm_last_error"]
    assert all("_master" in n or n in ("dpmm_step_stats_device", "dpmm_suffstats_device") for n in per["dpmm_hip_master.h"])
    assert len(per["dpmm_hip.h"]) <= 52
    for h in WORKER_HEADERS:      # each header compiles on its own as C
        import subprocess
        subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", h)])


def test_header_and_binding_agree(pkg):
    from dpmmsubclusters_jl_amd import binding
    assert declared_functions() == sorted(n for n, _, _ in binding.ABI)


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg.lib_path())
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.dpmm_abi_version.restype = ctypes.c_int
    assert lib.dpmm_abi_version() == 3


def test_host_library_exports_every_declared_symbol(pkg):
    """include/dpmm_host.h (the master half, libdpmmhost.so): every declared dpmmh_* function is exported; the worker table the
    Python binding builds has the layout of struct dpmmh_worker (ctx, rank, world, then the function pointers in header order)."""
    import importlib
    native = importlib.import_module("dpmmsubclusters_jl_amd.host.native")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    src = open(os.path.join(ROOT, "include", "dpmm_host.h")).read()
    body = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(dpmmh_[a-z0-9_]+)\s*\(", body)) - {"dpmmh_split_hook"})
    lib = ctypes.CDLL(native.build_library())
    for n in names:
        assert hasattr(lib, n), n
    lib.dpmmh_abi_version.restype = ctypes.c_int
    assert lib.dpmmh_abi_version() == 5
    struct = body[body.index("typedef struct dpmmh_worker {"):body.index("} dpmmh_worker;")]
    members = re.findall(r"\(\*([a-z_]+)\)\s*\(", struct)
    assert [f[0] for f in engine.WorkerTable._fields_] == ["ctx", "rank", "world"] + members
    # every member is bound to a libdpmmhip entry point that the worker header declares
    assert [m for m, _, _ in engine._NATIVE_MAP] == members
    assert set(sym for _, sym, _ in engine._NATIVE_MAP) <= set(declared_functions())


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.DpmmError) as e:
        pkg.Worker(pkg.PRIOR_NIW, 4, 10, device=0)
    assert e.value.code == -2  # DPMM_ENODEVICE


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "dpmmsubclusters.jl_amd")
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt and "orc_" not in txt, f


