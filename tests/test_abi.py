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


def test_reserved_register_check_of_the_lean_kernels():
    """niw_lean.hip touches the next tile's rows behind the compiler's back, in two flavours.  REGS (inline-asm loads into v254 /
    v255, kept free by amdgpu_num_vgpr(254)) is safe only while the allocator stays away: round 6's first change that added register pressure made
    it use them and this check -- run by the Makefile on the generated assembly -- stopped the build; the kernel with the direction screen therefore
    uses LDS (LDS-DMA loads without a register destination inside a save / restore of M0).  The check, per kernel: a kernel with register touches
    names v254 / v255 nowhere else; an LDS touch sits inside a well-formed M0 window."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("crv", os.path.join(ROOT, "dpmmsubclusters.jl_amd", "csrc", "check_reserved_vgprs.py"))
    crv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(crv)
    regs = ["\tglobal_load_dword v254, v[214:215], off", "\tglobal_load_dword v255, v[216:217], off", "\tv_add_f32 v253, v1, v2 ; v255 in a comment",
            "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[250:253], v[10:13], v[0:3]"]
    bad, reg_t, lds_t = crv.offending(regs)
    assert bad == [] and reg_t == 2 and lds_t == 0
    assert crv.offending(regs + ["\tv_mov_b32 v254, v3"])[0]
    assert crv.offending(regs + ["\tv_mfma_f32_16x16x32_bf16 v[0:3], v[252:255], v[10:13], v[0:3]"])[0]
    assert crv.offending(regs + ["\tglobal_load_dwordx2 v[254:255], v[2:3], off"])[0]
    assert crv.offending(regs + ["\tscratch_store_dword off, v255, s32"])[0]
    assert crv.offending(["\tv_mov_b32 v254, v3", "\tds_bpermute_b32 v255, v240, v105"])[0] == []      # a kernel WITHOUT register touches may use every register
    lds = ["\ts_mov_b32 s36, m0", "\ts_mov_b32 m0, s73", "\ts_nop 0", "\tglobal_load_lds_dword v[214:215], off", "\tglobal_load_lds_dword v[216:217], off offset:256",
           "\ts_mov_b32 m0, s36", "\tv_mov_b32 v255, v2"]
    bad, reg_t, lds_t = crv.offending(lds)
    assert bad == [] and reg_t == 0 and lds_t == 2
    assert crv.offending(["\tglobal_load_lds_dword v[2:3], off"])[0]                                   # no window
    assert crv.offending(lds[:3] + ["\tv_readlane_b32 s4, v3, 2"] + lds[3:])[0]                        # a foreign instruction inside the window
    assert crv.offending(lds[:5])[0]                                                                    # never restored
    assert crv.offending(lds[:5] + ["\ts_mov_b32 m0, s37"])[0]                                         # restored from another register
    built = os.path.join(ROOT, "dpmmsubclusters.jl_amd", "csrc", "build", "niw_lean.s")
    if os.path.exists(built):       # the assembly of THIS build, when the library was built here: every kernel of it
        ks = crv.kernels(open(built).read().split("\n"))
        assert len(ks) >= 3
        seen = 0
        for name, start, body in ks:
            bad, reg_t, lds_t = crv.offending(body, start + 1)
            assert bad == [], (name, bad[:3])
            seen += reg_t + lds_t
        assert seen >= 6
