# Runs scripts/config_step.py (C4: Multinomial D=1000 N=1e6) against the -DDPMM_U8_STAMPS build of the library (bash scripts/build_variant.sh u8stamps
# -DDPMM_U8_STAMPS): three workgroups print their cycle totals per phase of mult_sweep_u8_kernel (setup / feature passes / label draw / sub-label draw).
import importlib, sys, os
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
alt = os.path.abspath("dpmmsubclusters.jl_amd/lib/libdpmmhip_u8stamps.so")
b.lib_path = lambda: alt
sys.argv = ["config_step.py", "mult", "1000", "1000000", "3"]
exec(open("scripts/config_step.py").read())
