import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from one_to_many_gan_amd import dist as od

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
cfg = bench.make_config(64, 3, 2)
tr = bench.Trainer(bench.product_namespace("bf16"), cfg, dev, seed_offset=rank)
opts = [tr.oD, tr.oG, tr.oM, tr.oS]
od.broadcast_parameters(opts)
reds = [od.BucketReducer(o) for o in opts]
import traceback, collections
fires = collections.Counter()
names = {id(p): f"{n}.{k}" for n, net in zip("DGMS", (tr.D, tr.G, tr.M, tr.S)) for k, p in net.named_parameters()}
for r in reds:
    orig_on = r._on_grad
    def wrapped(param, _o=orig_on):
        st = traceback.extract_stack(limit=4)
        fires[(names.get(id(param), "?"), st[-2].name)] += 1
        return _o(param)
    r._on_grad = wrapped
    from one_to_many_gan_amd import ops
    for p_ in r.bucket.params:
        ops.GRAD_READY_HOOKS[p_] = wrapped
tr.kl_hook = od.make_kl_moment_hook()

def chk(t):
    v = torch.tensor([float(t.double().sum())], dtype=torch.float64)
    out = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(out, v)
    return [float(o) for o in out]

for name, o in zip("DGMS", opts):
    if rank == 0: print("init", name, chk(o.bucket.flat))
    else: chk(o.bucket.flat)
# monkeypatch step to report
import one_to_many_gan_amd.optim as optim
orig = optim.FusedAdam.step
def step(self):
    r = [x for x in reds if x.opt is self][0]
    info = (r.left, r.pending, r.launched)
    for h in self.pre_step_hooks: h()
    torch.cuda.synchronize()
    g = chk(self.bucket.grad)
    if rank == 0: print("step", "DGMS"[opts.index(self)], "left/pending/launched before wait", info, "grad sums", g, flush=True)
    self.pre_step_hooks, saved = [], self.pre_step_hooks
    orig(self)
    self.pre_step_hooks = saved
optim.FusedAdam.step = step
tr.step()
if rank == 0:
    for k, v in sorted(fires.items()):
        if k[0].startswith("D."): print(k, v)
for name, o in zip("DGMS", opts):
    c = chk(o.bucket.flat)
    if rank == 0: print("final", name, c)
dist.destroy_process_group()
