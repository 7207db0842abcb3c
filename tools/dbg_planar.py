"""plane-major vs channels-last input of a 3x3x3 convolution (debug helper).  GPU box only."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L, ops as O, plan as P
DEV = "cuda:0"
for cin, cout, d, B in ((256, 64, 8, 1), (256, 64, 8, 3), (256, 64, 8, 4), (256, 64, 10, 2), (256, 64, 12, 2), (256, 64, 16, 2), (512, 64, 8, 2), (272, 64, 8, 2)):
    dims = (d, d, d)
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(B, cin, *dims, generator=g).bfloat16().float()
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / math.sqrt(27 * cin)
    op = P.conv_fwd_op(cin, cout, 3, 1, 0, dims, cin, cout, L.SP_BF16)
    run = O.ConvRunner(op, DEV)
    run.prep(w.to(DEV), torch.zeros(cout, device=DEV))
    xs = O.alloc_cl(B, dims, cin, L.SP_BF16, DEV)
    O.ncdhw_to_cl(x.to(DEV), xs, L.SP_BF16)
    xp = xs.view(B, *dims, cin // 16, 16).permute(4, 0, 1, 2, 3, 5).contiguous().view(B, *dims, cin)
    ref = F.conv3d(x, w.bfloat16().float())
    for planar, xin in ((False, xs), (True, xp)):
        y = O.alloc_cl(B, op.y_dims, cout, L.SP_BF16, DEV)
        run.run(xin, y, B, None, None, L.ACT_NONE, 0.0, None, x_planar=planar)
        out = torch.empty((B, cout) + tuple(op.y_dims), device=DEV)
        O.cl_to_ncdhw(y, out, L.SP_BF16)
        t = op.subs[0].tile
        print("%d->%d @%d planar=%s  max err %.3g  nan=%s  tile MT=%d ngroups=%d opg=%d dma=%d nt=%d" % (
            cin, cout, d, planar, float((out.cpu() - ref).abs().max()), bool(torch.isnan(out).any()), t["MT"], t["ngroups"], t["octs_per_group"], t["dma"], op.nt))
