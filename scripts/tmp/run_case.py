import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_multirank as t
case = sys.argv[1]
out = "/tmp/o_%s.npz" % case
t._run(0, 1, 29990, out, case, "gloo")
d = np.load(out)
print(case, "K", d["K"].tolist(), "nmi", d["nmi"][-1])
