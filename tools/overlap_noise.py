import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import run_case
from tests.namespaces import product_ns
from one_to_many_gan_amd import ops
def run(flag):
    ops._D_OVERLAP = flag
    return run_case("steps64", product_ns("fp32"), "cuda")
def err(a, b):
    return max(float((a[k].double() - b[k].double()).norm() / max(float(b[k].double().norm()), 1e-30)) for k in a)
a, b = run(False), run(False)
c, d = run(True), run(True)
print("off vs off", err(a, b)); print("on vs on", err(c, d)); print("off vs on", err(a, c), err(b, d))
ops._WGRAD_STREAM = False; ops._GROUP_STREAM = False
e, f = run(False), run(False)
print("single stream: off vs off", err(e, f), "; vs multi-stream off", err(a, e))
