"""Per-parameter gradient differences of the exact 2-rank mode vs the single-process whole-batch run (diagnostic)."""
import os, sys
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CH = [2, 16, 32, 64, 32, 16, 32, 2]


def _run(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stroke_prediction_amd  # noqa
    from oracle import weights as W
    from stroke_prediction_amd.common.model.Unet3D import Unet3D
    from stroke_prediction_amd.common.metrics import BatchDiceLoss
    import stroke_prediction_amd.common.dto.UnetDto as UD
    from stroke_prediction_amd.optim import attach_flat_grads
    from stroke_prediction_amd.parallel import DataParallelSync
    dev = "cuda:0"
    x, y = W.unet_inputs(4, (44, 44, 44), 31)
    crit = BatchDiceLoss([1.0])

    def run(model, xs, ys):
        attach_flat_grads(model)
        dto = model(UD.init_dto(xs.to(dev), ys[:, 0:1].to(dev), ys[:, 1:2].to(dev)))
        loss = (crit(dto.outputs.core, dto.given_variables.core) + crit(dto.outputs.penu, dto.given_variables.penu)) / 2
        loss.backward()
        return {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}

    def fresh():
        m = Unet3D(CH, dtype="f32")
        m.load_state_dict(W.make_state_dict(W.unet_spec(CH), 31))
        return m.to(dev).train()

    ref = run(fresh(), x, y) if rank == 0 else None
    ref2 = run(fresh(), x, y) if rank == 0 else None
    dist.barrier()
    model = fresh()
    sync = DataParallelSync(model, mode="exact")
    g = run(model, x[rank * 2:rank * 2 + 2], y[rank * 2:rank * 2 + 2])
    sync.close()
    if rank == 0:
        out = []
        for n in ref:
            d = float((g[n] - ref[n]).norm() / (ref[n].norm() + 1e-30))
            d2 = float((ref2[n] - ref[n]).norm() / (ref[n].norm() + 1e-30))
            out.append((n, d, d2, float(ref[n].norm())))
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, 2, 29751, q)) for r in range(2)]
    for p in procs: p.start()
    res = q.get(timeout=600)
    for p in procs: p.join(timeout=120)
    for n, d, d2, nr in res:
        print("%-45s exact-vs-single %.3e   single-vs-single %.3e   |g| %.3e" % (n, d, d2, nr))
