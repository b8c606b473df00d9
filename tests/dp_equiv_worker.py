"""Worker of tests/test_dist_gpu.py::test_data_parallel_step_equals_the_big_batch_step.

mode "dp": one rank of a 2-rank run (both ranks on GPU 0, gloo), local batch 2; mode "single": one process, batch 4.
Both run ONE discriminator_step + ONE generator_step of the product on the SAME four samples (rank r holds samples
2 r, 2 r + 1) with every random draw replaced by a closed-form table indexed by the GLOBAL sample (latents z, theta, h;
style mixing off), and dump the gradient every optimiser is about to consume -- for "dp" after the bucket all-reduce,
scaled by the 1/world the fused Adam applies.  Data parallelism is exact for this step (per-sample networks, batch-mean
losses, global-batch KL moments through the hook): the two dumps must agree to rounding."""
import contextlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GLOBAL_B = 4


@contextlib.contextmanager
def tabled_draws(lo, hi):
    """torch.randn / torch.rand / Tensor.uniform_ from tables keyed by (call number, global sample index)."""
    from oracle.detweights import sym_uniform, unit_uniform

    calls = {"randn": 0, "rand": 0, "uniform": 0}
    o_randn, o_rand, o_uniform = torch.randn, torch.rand, torch.Tensor.uniform_

    def randn(*shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if kw.get("device") is not None or len(shape) != 2:
            return o_randn(*shape, **kw)
        calls["randn"] += 1
        return (sym_uniform(f"dp/randn{calls['randn']}", (GLOBAL_B, shape[1])) * 1.7)[lo:hi].clone()

    def rand(*shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if kw.get("device") is not None:
            return o_rand(*shape, **kw)
        if shape == ():
            return torch.ones(())  # the style-mixing decision: never below the probability
        calls["rand"] += 1
        return unit_uniform(f"dp/rand{calls['rand']}", (GLOBAL_B,))[lo:hi].clone()

    def uniform_(self, a=0.0, b=1.0, **kw):
        if self.dim() != 1 or self.numel() != hi - lo:
            return o_uniform(self, a, b, **kw)
        calls["uniform"] += 1
        t = unit_uniform(f"dp/unif{calls['uniform']}", (GLOBAL_B,))[lo:hi] * (b - a) + a
        return self.copy_(t.to(self.device))

    torch.randn, torch.rand, torch.Tensor.uniform_ = randn, rand, uniform_
    try:
        yield
    finally:
        torch.randn, torch.rand, torch.Tensor.uniform_ = o_randn, o_rand, o_uniform


def main(out_dir, mode):
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.core import training as pt
    from one_to_many_gan_amd.model import loss as plo
    from tests.cases import build_step_state, image_batch, make_config
    from tests.namespaces import product_ns

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rank, world = 0, 1
    if mode == "dp":
        import torch.distributed as dist

        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
    local = GLOBAL_B // world
    lo, hi = rank * local, (rank + 1) * local
    ns = product_ns("fp32")  # the parity mode: differences are summation order only
    cfg = make_config(1, (64, 64), local)
    cfg["training"]["style_mixing_prob"] = 0.0
    nets, opts = build_step_state(ns, dev, cfg, "dp")
    kl_hook = None
    ada_p = plo.ADAp(256, 5.12e-4, local, 0.6)
    if mode == "dp":
        from one_to_many_gan_amd import dist as o2m_dist

        o2m_dist.broadcast_parameters(opts.values())
        reducers = {k: o2m_dist.BucketReducer(o) for k, o in opts.items()}
        kl_hook = o2m_dist.make_kl_moment_hook()
        o2m_dist.sync_ada_p(ada_p)
    grads = {}
    for k, o in opts.items():
        o.pre_step_hooks.append(lambda k=k, o=o: grads.__setitem__(k, (o.bucket.grad * o.grad_scale).clone()))

    def batches(stream):
        i = 0
        while True:
            yield image_batch(f"dp/{stream}{i}", (GLOBAL_B, 1, 64, 64))[lo:hi].to(dev)
            i += 1

    prints, marks = batches("print"), batches("mark")
    buf = pt.ImageBuffer(100)
    ada = o2m.IdentityADA().to(dev)
    with tabled_draws(lo, hi):
        d_out = pt.discriminator_step(cfg, dev, nets["D"], nets["G"], nets["M"], opts["D"], prints, marks, buf, ada, ada_p)
        kw = {"kl_moment_hook": kl_hook} if kl_hook is not None else {}
        g_out = pt.generator_step(cfg, dev, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["M"], opts["S"],
                                  prints, marks, ada, **kw)
    torch.cuda.synchronize()
    out = {k: v.double().cpu() for k, v in grads.items()}
    out["kl"] = torch.tensor(float(g_out[1][3]), dtype=torch.float64)
    if mode == "dp":
        out["launch_logs"] = {k: [why for _, why in r.last_launch_log] for k, r in reducers.items()}
    torch.save(out, os.path.join(out_dir, f"{mode}{rank}.pt"))
    if mode == "dp":
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
