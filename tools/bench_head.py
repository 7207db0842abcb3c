"""Time sp_head_fwd / sp_head_bwd alone at the headline shape (B=4, 88^3, C=16, CH=32, NC=2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stroke_prediction_amd.runtime import lib as L, ops as O

B, n, C, CH, NC = 4, 88, 16, 32, 2
nv = n ** 3
dev = "cuda:0"
x = torch.randn(B, n, n, n, C, device=dev).bfloat16()
w1 = torch.randn(CH, C, device=dev) * 0.2; b1 = torch.randn(CH, device=dev) * 0.1
w2 = torch.randn(NC, CH, device=dev) * 0.2; b2 = torch.randn(NC, device=dev) * 0.1
seg = torch.empty(B, NC, n, n, n, device=dev); dseg = torch.randn_like(seg)
dz = torch.empty_like(x); dbs = torch.zeros(C, dtype=torch.float64, device=dev)
lib = L.load()
rows, nq = lib.sp_head_bwd_rows(B * nv), lib.sp_head_row_floats(C, CH, NC)
part = torch.empty(rows * nq, device=dev)
g1 = torch.zeros(CH * C, device=dev); g2 = torch.zeros(CH, device=dev); g3 = torch.zeros(NC * CH, device=dev); g4 = torch.zeros(NC, device=dev)

def fwd():
    L.call("sp_head_fwd", O.ptr(x), L.SP_BF16, nv, B, C, C, O.ptr(w1), O.ptr(b1), CH, O.ptr(w2), O.ptr(b2), NC, 0.01, O.ptr(seg), O.stream())
def bwd():
    L.call("sp_head_bwd", O.ptr(x), L.SP_BF16, nv, B, C, C, O.ptr(w1), O.ptr(b1), CH, O.ptr(w2), NC, 0.01, O.ptr(seg), O.ptr(dseg),
           L.ACT_LEAKY, 0.01, O.ptr(dz), O.ptr(part), O.stream())
def fin():
    L.call("sp_head_grad_finish", O.ptr(part), rows, C, CH, NC, O.ptr(g1), O.ptr(g2), O.ptr(g3), O.ptr(g4), O.ptr(dbs), O.stream())
for name, f in (("fwd", fwd), ("bwd", bwd), ("finish", fin)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("head %s: %.1f us" % (name, e0.elapsed_time(e1) * 1000 / 20))
