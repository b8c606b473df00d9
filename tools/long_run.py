"""Stability check: N D+G steps of train.run on synthetic data at the stock hyper-parameters (256x256x3, batch 16, bf16),
printing the reference's log line every 50 steps -- finite, slowly moving losses expected (no dataset: the numbers mean
nothing beyond that).  Usage: long_run.py [steps] [size] [batch] [--graph]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import one_to_many_gan_amd as o2m
import train
from pathlib import Path
from tests.cases import make_config

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
graph = "--graph" in sys.argv
dev = torch.device("cuda:0")
o2m.set_precision(os.environ.get("LONG_RUN_PRECISION", "bf16"))
cfg = make_config(3 if size > 64 else 1, (size, size), batch)
cfg["training"].update(checkpoint_directory=Path(tempfile.mkdtemp()), training_run="long", training_steps=steps)
cfg["evaluation"] = {"log_interval": 50, "checkpoint_interval": 10 ** 9, "n_evaluation_images": 0, "inference_batch_size": 2}
lines = []
def log(l):
    lines.append(l); print(l, flush=True)
nets, opts = train.run(cfg, dev, steps, train.synthetic_batches(10, cfg, dev), train.synthetic_batches(20, cfg, dev),
                       log=log, image_grids=False, graph=graph)
torch.cuda.synchronize()
ok = all(bool(torch.isfinite(o.bucket.flat).all()) for o in opts.values())
print("parameters finite:", ok, "| log lines with nan:", sum("nan" in l.lower() for l in lines if l.startswith("Step")))
