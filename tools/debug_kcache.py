"""Diagnostic: does a kernel that reads weights through the scalar cache see in-place updates made by earlier kernels?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from stroke_prediction_amd.runtime import lib as L, ops as O
dev = "cuda:0"
B, n, C, CH, NC = 1, 8, 16, 32, 2
nv = n ** 3
x = torch.randn(B, n, n, n, C, device=dev).bfloat16()
w1 = torch.randn(CH, C, device=dev) * 0.2; b1 = torch.randn(CH, device=dev) * 0.1
w2 = torch.randn(NC, CH, device=dev) * 0.2; b2 = torch.randn(NC, device=dev) * 0.1
seg = torch.empty(B, NC, n, n, n, device=dev)
def fwd():
    L.call("sp_head_fwd", O.ptr(x), L.SP_BF16, nv, B, C, C, O.ptr(w1), O.ptr(b1), CH, O.ptr(w2), O.ptr(b2), NC, 0.01, O.ptr(seg), O.stream())
def ref():
    xf = x.float().view(-1, C)
    h = torch.nn.functional.leaky_relu(xf @ w1.t() + b1, 0.01)
    return torch.sigmoid(h @ w2.t() + b2).t().reshape(B, NC, n, n, n)
bad = 0
for it in range(200):
    w1.add_(torch.randn_like(w1) * 0.05); b1.add_(0.01); w2.mul_(1.01); b2.add_(0.003)     # in-place updates by other kernels
    fwd()
    err = float((seg - ref()).abs().max())
    if err > 1e-4:
        bad += 1
        if bad < 5: print("iter", it, "max err", err)
print("stale results: %d / 200" % bad)
