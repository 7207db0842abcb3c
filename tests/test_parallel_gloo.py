"""Data-parallel gradient exchange with world_size 2 on the CPU (gloo): one all-reduce of the flat gradient
buffer, parameters broadcast from rank 0, FusedAdam's grad_scale = 1/world turning the SUM into a mean."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stroke_prediction_amd  # noqa: F401
    from stroke_prediction_amd.common.model.Unet3D import Unet3D
    from stroke_prediction_amd.optim import attach_flat_grads
    from stroke_prediction_amd.parallel import DataParallelSync
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    model = Unet3D([2, 16, 32, 64, 32, 16, 32, 2])
    sync = DataParallelSync(model)
    flat_p, flat_g = model.flat_buffers()
    gathered = [torch.zeros_like(flat_p) for _ in range(world)]
    dist.all_gather(gathered, flat_p)
    same_params = all(torch.equal(gathered[0], g) for g in gathered)
    attach_flat_grads(model)
    flat_g.fill_(float(rank + 1))
    model._after_backward()                            # what the autograd node calls after filling the gradients
    ok_sum = bool(torch.all(flat_g == sum(range(1, world + 1))))
    views_alias = all(p.grad.data_ptr() >= flat_g.data_ptr() for p in model.parameters())
    q.put((rank, same_params, ok_sum, views_alias, sync.grad_scale, flat_g.numel()))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    world, port = 2, 29731
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_params, ok_sum, views_alias, scale, n in res:
        assert same_params and ok_sum and views_alias
        assert scale == 0.5 and n == 355014
