"""Two training steps at the reference's stock geometry (config.toml: 512x256, 1 channel), batch 4."""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import train
import one_to_many_gan_amd as o2m
from tests.cases import make_config

o2m.set_precision("bf16")
cfg = make_config(1, (512, 256), 4)
cfg["training"].update(checkpoint_directory=tempfile.mkdtemp(), training_run="stock", training_steps=3)
cfg["evaluation"] = {"log_interval": 1, "checkpoint_interval": 1000, "n_evaluation_images": 0, "inference_batch_size": 4}
dev = torch.device("cuda:0")
lines = []
train.run(cfg, dev, 3, train.synthetic_batches(1, cfg, dev), train.synthetic_batches(2, cfg, dev), log=lines.append)
print("\n".join(lines))
assert all("nan" not in l.lower() for l in lines)
print("stock-shape run OK")
