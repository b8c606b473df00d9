"""World-size-2 CPU (gloo) tests of the data-parallel machinery in
one_to_many_gan_amd/dist.py: the segmented flat-bucket gradient all-reduce launched from inside
backward, parameter broadcast, the global-batch KL moment hook and the ADAp score average.  The
HIP kernels are not involved (no GPU here): gradients are produced by plain torch ops on the same
FlatBucket / PreparedWeight / hook plumbing the GPU path uses (the one kernel that plumbing calls,
``wgrad_finalize``, is stood in for by its formula)."""

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


class _FilterNet(torch.nn.Module):
    """Registration order stem -> mid -> head; backward reaches head first.  ``mid`` and ``head``
    are "conv filters" (4-D): their gradients take the ops.PreparedWeight route (kernel-layout
    accumulators, finalised per layer) like the real convolutions; ``mid`` is applied twice."""

    def __init__(self):
        super().__init__()
        lin = torch.nn.Linear(6, 8)  # direct parameters: a module's own come before its children's
        self.stem_w, self.stem_b = torch.nn.Parameter(lin.weight.detach().clone()), torch.nn.Parameter(lin.bias.detach().clone())
        self.mid = torch.nn.Parameter(torch.randn(8, 8, 1, 1))
        self.head = torch.nn.Parameter(torch.randn(8, 8, 1, 1))
        self.preps = None

    def stem(self, x):
        return x @ self.stem_w.t() + self.stem_b

    def forward(self, x):
        h = torch.tanh(self.stem(x))
        h = torch.tanh(_filter_op(h, self.mid, self.preps["mid"]))
        h = torch.tanh(_filter_op(h, self.mid, self.preps["mid"]))
        return _filter_op(h, self.head, self.preps["head"])


class _FilterFn(torch.autograd.Function):
    """y = x @ W^T with the weight gradient accumulated the way ops._ConvFn does it."""

    @staticmethod
    def forward(ctx, x, weight, prep, fail):
        ctx.prep, ctx.fail = prep, fail
        ctx.counted = prep.note_forward_use(ctx.needs_input_grad[1])
        ctx.save_for_backward(x, weight)
        return x @ weight[:, :, 0, 0].t()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        dw_acc, _ = ctx.prep.accumulators(g.device)
        dw_acc[:, 0, 0, :] += g.t() @ x  # kernel layout [Co][KH][KW][Ci]
        if ctx.fail:
            raise RuntimeError("injected failure inside backward")
        if ctx.counted:
            ctx.prep.use_reduced(g.device)
        return g @ weight[:, :, 0, 0], None, None, None


def _filter_op(x, weight, prep, fail=False):
    return _FilterFn.apply(x, weight, prep, fail)


def _cpu_wgrad_finalize(acc, gq, w32, grad, co, ci, c):
    """Formula of o2m_wgrad_finalize (include/o2m_hip.h) for gq = None."""
    grad += c * acc.permute(0, 3, 1, 2)[:co, :ci]
    acc.zero_()


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
            assert [why for _, why in reducer.launch_log] == (["hook"] if step == 1 else [])
            reducer.launch_log.clear()
            local = opt.bucket.grad.clone()  # may already be reduced: recompute the local grad
            ref_net = _Net()
            ref_net.load_state_dict(net.state_dict())
            (ref_net(x).square().mean() + extra(ref_net)).backward()
            mine = torch.cat([
                (p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in ref_net.parameters()])
            for hook in opt.pre_step_hooks:
                hook()
            # wait() closes the step: the log moves to last_launch_log and the live one starts empty
            assert [why for _, why in reducer.last_launch_log] == (["wait"] if step == 0 else [])
            assert reducer.launch_log == []
            # gather every rank's local gradient and compare with the reduced bucket
            allg = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allg, mine)
            want_sum = torch.stack(allg).sum(0)
            got = torch.cat([opt.bucket.grad[o:o + p.numel()] for p, o in zip(opt.bucket.params, opt.bucket.offsets)])
            assert torch.allclose(got, want_sum, atol=1e-6), (step, (got - want_sum).abs().max())
            results.append(float(got.abs().sum()))
            del local

        _segments_launch_inside_backward(rank, world)

        # ADAp: every rank's controller sees the mean confidence over ranks
        from one_to_many_gan_amd.model.loss import ADAp

        ctl = o2m_dist.sync_ada_p(ADAp(8, 0.05, 4, 0.6))  # window of 2 calls, step 0.4
        for k in range(3):
            ctl.update_p(torch.tensor(1.0 if rank == 0 else 0.4))  # mean 0.7 > 0.6 on BOTH ranks
        assert abs(ctl() - 0.4) < 1e-6 and all(abs(float(v) - 0.7) < 1e-6 for v in ctl.mean_real_scores)

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


def _segments_launch_inside_backward(rank, world):
    """Conv-filter route: a segment whose filters completed early is all-reduced from INSIDE
    backward (before the end-of-backward callback), last segment first; a backward that raises
    does not poison the next one."""
    from one_to_many_gan_amd import _hip, ops
    from one_to_many_gan_amd import dist as o2m_dist
    from one_to_many_gan_amd import optim

    _hip.wgrad_finalize = _cpu_wgrad_finalize
    torch.manual_seed(5)
    net = _FilterNet()
    opt = optim.FusedAdam.__new__(optim.FusedAdam)
    opt.bucket = optim.FlatBucket(net)
    opt.lr, opt.betas, opt.eps, opt.grad_scale, opt.pre_step_hooks = 1e-2, (0.5, 0.99), 1e-8, 1.0, []
    net.preps = {"mid": ops.PreparedWeight(net.mid, False), "head": ops.PreparedWeight(net.head, False)}
    for prep in net.preps.values():
        prep.get = lambda: (None, None, None, None, None)  # w32 is only read for modulated layers
        prep.c = 1.0
    reducer = o2m_dist.BucketReducer(opt, segment_bytes=4 * 56)  # stem (w+b = 56 floats) | mid | head
    assert reducer.n_seg == 3 and [n for _, _, n in reducer.seg_range] == [2, 1, 1]
    events = []
    orig_cb, orig_launch = ops._finalize_weight_grads, reducer._launch
    reducer._launch = lambda seg, why: (events.append(("launch", seg, why)), orig_launch(seg, why))[1]
    ops._finalize_weight_grads = lambda: (events.append(("end_of_backward",)), orig_cb())[1]
    try:
        for step in range(2):
            opt.bucket.zero_grad()
            torch.manual_seed(50 + rank + 10 * step)
            x = torch.randn(4, 6)
            net(x).square().mean().backward()
            # head (seg 2), then mid after its SECOND use (seg 1), then the stem's weight + bias
            # (seg 0): all three from hooks, none left for wait(); the end-of-backward callback runs after
            assert events == [("launch", 2, "hook"), ("launch", 1, "hook"), ("launch", 0, "hook"),
                              ("end_of_backward",)], events
            events.clear()
            ref = _FilterNet()
            ref.load_state_dict(net.state_dict())
            h = torch.tanh(ref.stem(x))
            w_mid, w_head = ref.mid[:, :, 0, 0], ref.head[:, :, 0, 0]
            (torch.tanh(torch.tanh(h @ w_mid.t()) @ w_mid.t()) @ w_head.t()).square().mean().backward()
            mine = torch.cat([p.grad.flatten() for p in ref.parameters()])
            reducer.wait()
            allg = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allg, mine)
            got = torch.cat([opt.bucket.grad[o:o + p.numel()] for p, o in zip(opt.bucket.params, opt.bucket.offsets)])
            assert torch.allclose(got, torch.stack(allg).sum(0), atol=1e-5), (got - torch.stack(allg).sum(0)).abs().max()

        # a backward that dies after a filter was touched must not disable later passes
        reducer.enabled = False
        opt.bucket.zero_grad()
        x = torch.randn(4, 6)
        y = _filter_op(torch.tanh(_filter_op(torch.tanh(net.stem(x)), net.mid, net.preps["mid"])), net.head,
                       net.preps["head"], True)
        with pytest.raises(RuntimeError, match="injected"):
            y.sum().backward()
        assert ops._PENDING and net.preps["head"].pending  # what round 1 was left with forever
        net(x).square().mean().backward()
        assert not ops._PENDING and not any(p.pending for p in net.preps.values())
        ref = _FilterNet()
        ref.load_state_dict(net.state_dict())
        h = torch.tanh(ref.stem(x))
        w_mid, w_head = ref.mid[:, :, 0, 0], ref.head[:, :, 0, 0]
        (torch.tanh(torch.tanh(h @ w_mid.t()) @ w_mid.t()) @ w_head.t()).square().mean().backward()
        for a, b in zip(net.parameters(), ref.parameters()):  # the aborted pass left nothing behind
            assert torch.allclose(a.grad, b.grad, atol=1e-6)
    finally:
        ops._finalize_weight_grads = orig_cb


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
