"""World-size-2 CPU (gloo) tests of the data-parallel machinery in
one_to_many_gan_amd/dist.py: the flat-bucket gradient all-reduce launched from autograd
hooks, parameter broadcast, and the global-batch KL moment hook.  The HIP kernels are not
involved (no GPU here): gradients are produced by plain torch ops on the same FlatBucket /
hook plumbing the GPU path uses, and the fused Adam launch is replaced by its reference
formula so that the averaging factor (grad_scale = 1/N) is exercised too."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(5, 7)
        self.b = torch.nn.Linear(7, 3, bias=False)
        self.unused = torch.nn.Parameter(torch.zeros(4))  # never gets a gradient

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from one_to_many_gan_amd import dist as o2m_dist
        from one_to_many_gan_amd import optim

        torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
        net = _Net()
        opt = optim.FusedAdam.__new__(optim.FusedAdam)  # build without touching the GPU library
        opt.bucket = optim.FlatBucket(net)
        opt.lr, opt.betas, opt.eps, opt.grad_scale, opt.pre_step_hooks = 1e-2, (0.5, 0.99), 1e-8, 1.0, []
        o2m_dist.broadcast_parameters([opt])
        flat0 = opt.bucket.flat.clone()
        gathered = [torch.empty_like(flat0) for _ in range(world)]
        dist.all_gather(gathered, flat0)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "broadcast did not equalise weights"

        reducer = o2m_dist.BucketReducer(opt)
        assert opt.grad_scale == 1.0 / world

        results = []
        for step in range(2):  # two steps: the reducer must re-arm itself
            opt.bucket.zero_grad()
            torch.manual_seed(7 + 10 * step + rank)
            x = torch.randn(6, 5)
            # step 0: `unused` gets no gradient -> the reducer launches from wait();
            # step 1: every parameter fires its hook -> the all-reduce is launched from autograd
            extra = (lambda n: n.unused.sum()) if step == 1 else (lambda n: 0.0)
            (net(x).square().mean() + extra(net)).backward()
            assert reducer.launched == (step == 1)
            local = opt.bucket.grad.clone()  # may already be reduced: recompute the local grad
            ref_net = _Net()
            ref_net.load_state_dict(net.state_dict())
            (ref_net(x).square().mean() + extra(ref_net)).backward()
            mine = torch.cat([
                (p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in ref_net.parameters()])
            for hook in opt.pre_step_hooks:
                hook()
            # gather every rank's local gradient and compare with the reduced bucket
            allg = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allg, mine)
            want_sum = torch.stack(allg).sum(0)
            got = torch.cat([opt.bucket.grad[o:o + p.numel()] for p, o in zip(opt.bucket.params, opt.bucket.offsets)])
            assert torch.allclose(got, want_sum, atol=1e-6), (step, (got - want_sum).abs().max())
            results.append(float(got.abs().sum()))
            del local

        # KL moment hook: global-batch moments + gradient scaled by world size
        hook = o2m_dist.make_kl_moment_hook()
        t = torch.full((4,), float(rank + 1), requires_grad=True)
        s1, s2, n = hook(t.sum(), t.square().sum(), t.numel())
        assert n == 4 * world
        assert float(s1) == sum(4.0 * (r + 1) for r in range(world))
        assert float(s2) == sum(4.0 * (r + 1) ** 2 for r in range(world))
        (s1 + s2).backward()
        assert torch.allclose(t.grad, world * (1 + 2 * t.detach()))
        q.put((rank, "ok", results))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, "fail", traceback.format_exc() + str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucket_allreduce_and_kl_hook_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status, payload in out:
        assert status == "ok", f"rank {rank}: {payload}"
    assert out[0][2] == out[1][2]  # both ranks hold the same reduced gradients
