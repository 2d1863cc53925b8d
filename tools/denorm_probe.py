"""Does the fp16 matrix pipe honour fp16 subnormal inputs on this GPU?  (It does on gfx950: the split operands of
lemon_linear_f16x3 / the attention kernel / the 16-bit kNN filter may contain them.)  python tools/denorm_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemon_amd import ops
a = torch.full((256, 256), 2.0 ** -20, dtype=torch.float16, device="cuda")
b = torch.full((256, 256), 2.0 ** 10, dtype=torch.float16, device="cuda")
print("fp16 subnormal x normal via matmul:", float((a @ b)[0, 0]), "expected", 256 * 2.0 ** -10)
x = torch.full((256, 256), 3.0e-6, dtype=torch.float32, device="cuda")      # hi is an fp16 subnormal
w = torch.full((256, 256), 1.0, dtype=torch.float32, device="cuda")
ws = ops.weight_scale_f16x3(w)
y = ops.linear_split(ops.split_operand(x, "f16x3"), ops.split_operand(w, "f16x3", weight=True, wscale=ws), alpha=1.0 / ws)
print("lemon_linear_f16x3 with subnormal-hi activations:", float(y[0, 0]), "expected", 256 * 3.0e-6)
