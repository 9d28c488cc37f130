"""Dev helper: does `import torch` still find the GPU after this library initialised HIP? (libamdhip64 load order, see binding.load_library)"""
import sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
import numpy as np
which = sys.argv[1]
if which == "baddev":
    try:
        pkg.Worker(pkg.PRIOR_NIW, 4, 10, device=99)
    except Exception as e:
        print("expected:", e)
elif which == "bigdim":
    try:
        pkg.Worker(pkg.PRIOR_NIW, 300, 10, device=0)
    except Exception as e:
        print("expected:", e)
elif which == "ok":
    wk = pkg.Worker(pkg.PRIOR_NIW, 4, 10, device=0); wk.close()
import torch
torch.cuda.set_device(0)
print(which, "torch ok", torch.cuda.device_count())
