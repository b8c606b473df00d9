"""Which small aten ops (outside torch.ops.o2m) one D+G step issues and from where: finds stray tiny launches.
TorchDispatchMode + the Python stack of each dispatch (ops issued by the C++ autograd engine show as <engine>)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

device = torch.device("cuda", 0)
torch.cuda.set_device(device)
cfg = bench.make_config(256, 3, 16)
tr = bench.Trainer(bench.product_namespace("bf16", 0.0), cfg, device, seed_offset=0)
tr.fixed_ada_p = 0.0
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = {"aten.empty.memory_format", "aten.as_strided.default", "aten.slice.Tensor", "aten.detach.default", "aten.view.default",
        "aten.empty_strided.default", "aten.empty_like.default", "aten.select.int", "aten.permute.default", "aten.t.default",
        "aten.transpose.int", "aten.alias.default", "aten.expand.default", "aten.unsqueeze.default", "aten.record_stream.default",
        "aten._unsafe_view.default", "aten.reshape.default", "aten.narrow.default", "aten.squeeze.dim", "aten.unbind.int",
        "aten.split.Tensor", "aten.chunk.default", "aten.is_pinned.default", "aten._local_scalar_dense.default"}
where = collections.Counter()

class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith("o2m.") and name not in SKIP:
            fr = "<engine>"
            for f in reversed(traceback.extract_stack(limit=14)):
                if f.filename.startswith(ROOT) and "aten_census" not in f.filename:
                    fr = f"{os.path.relpath(f.filename, ROOT)}:{f.lineno} {f.name}"
                    break
            if fr == "<engine>":
                node = torch._C._current_autograd_node()
                if node is not None:
                    fr = f"<engine: {node.name()}>"
            where[(name, fr)] += 1
        return func(*args, **(kwargs or {}))

with Census():
    tr.step()
torch.cuda.synchronize()
for (name, fr), c in where.most_common(160):
    print(f"{c:5d}  {name:34s} {fr}")
print("total", sum(where.values()))
