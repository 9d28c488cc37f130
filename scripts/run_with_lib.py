"""Dev helper: run pytest against an alternative build of the HIP library: python scripts/run_with_lib.py <lib.so> <pytest args...>"""
import sys, os, importlib
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
alt = os.path.abspath(sys.argv[1])
b.lib_path = lambda: alt
import pytest
sys.exit(pytest.main(sys.argv[2:]))
