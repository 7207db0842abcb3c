"""probe: pair upsample + crop + concat at a given shape against float64 (argv: B D CPu Ds CPs planar)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import stroke_prediction_amd
from stroke_prediction_amd.runtime import lib as L, ops as O
DEV = "cuda:0"
B, D, CPu, Ds, CPs, planar = [int(v) for v in sys.argv[1:7]]
g = torch.Generator(device=DEV).manual_seed(1)
def pair(shape):
    v = torch.randn(shape, generator=g, device=DEV)
    p = torch.empty((2,) + tuple(shape), dtype=torch.bfloat16, device=DEV)
    p[0] = v.bfloat16(); p[1] = (v - p[0].float()).bfloat16()
    return p[0], p[1]
lo_h, lo_l = pair((B, D, D, D, CPu))
sk_h, sk_l = pair((B, Ds, Ds, Ds, CPs))
cp = torch.full((2, B, 2 * D, 2 * D, 2 * D, CPu + CPs), 7.0, dtype=torch.bfloat16, device=DEV)
c_h, c_l = cp[0], cp[1]
cst = torch.zeros(L.SP_REDUCE_ROWS * (CPu + CPs) * 2, dtype=torch.float64, device=DEV)
lod = lambda a, c: c.data_ptr() - a.data_ptr()
print("deltas", lod(lo_h, lo_l), lod(sk_h, sk_l), lod(c_h, c_l), flush=True)
L.call("sp_upsample2_crop_cat_fwd_hl", O.ptr(lo_h), lod(lo_h, lo_l), CPu, O.ptr(sk_h), lod(sk_h, sk_l), CPs, O.ptr(c_h), lod(c_h, c_l), CPu + CPs,
       B, D, D, D, Ds, Ds, Ds, (B * 8 * D ** 3 * 16) if planar else 0, O.ptr(cst), O.stream())
torch.cuda.synchronize()
print("ran", flush=True)
lv = (lo_h.double() + lo_l.double()).permute(0, 4, 1, 2, 3)
up = F.interpolate(lv, scale_factor=2, mode="trilinear", align_corners=False)
o = (Ds - 2 * D) // 2
sv = (sk_h.double() + sk_l.double()).permute(0, 4, 1, 2, 3)[:, :, o:o + 2 * D, o:o + 2 * D, o:o + 2 * D]
ref = torch.cat((up, sv), 1)
cv = c_h.double() + c_l.double()
C = CPu + CPs
cv = cv.view(C // 16, B, 2 * D, 2 * D, 2 * D, 16).permute(1, 0, 5, 2, 3, 4).reshape(B, C, 2 * D, 2 * D, 2 * D) if planar else cv.permute(0, 4, 1, 2, 3)
print("max err", float((cv - ref).abs().max()), "ref max", float(ref.abs().max()))
