"""Worker of tests/test_dist_gpu.py::test_train_loop_under_data_parallelism: one rank of a 2-rank run of train.run
(both ranks on GPU 0, gloo) -- steps, rank-0 checkpoints, resume -- writing a checksum of its weights."""
import json
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir):
    import one_to_many_gan_amd as o2m
    import train
    from tests.cases import make_config

    rank = int(os.environ["RANK"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    o2m.set_precision("bf16")
    cfg = make_config(1, (64, 64), 2)
    run_dir = Path(out_dir)
    cfg["training"].update(checkpoint_directory=run_dir, training_run="dp", training_steps=2)
    cfg["evaluation"] = {"log_interval": 1, "checkpoint_interval": 1, "n_evaluation_images": 0, "inference_batch_size": 2}
    lines = []
    nets, opts = train.run(cfg, dev, 2, train.synthetic_batches(10 + rank, cfg, dev), train.synthetic_batches(20 + rank, cfg, dev),
                           log=lines.append, data_parallel=True, image_grids=False)
    torch.cuda.synchronize()
    sums = {k: float(o.bucket.flat.double().sum()) for k, o in opts.items()}
    dist.barrier()
    ck = run_dir / "dp" / "models" / "2.tar"
    # every rank resumes from rank 0's checkpoint and takes one more step
    nets2, opts2 = train.run(cfg, dev, 3, train.synthetic_batches(30 + rank, cfg, dev), train.synthetic_batches(40 + rank, cfg, dev),
                             resume=ck, log=lines.append, data_parallel=True, image_grids=False)
    torch.cuda.synchronize()
    sums2 = {k: float(o.bucket.flat.double().sum()) for k, o in opts2.items()}
    # how each bucket segment's all-reduce of the last step was launched: "hook" = from inside backward
    logs = {k: [why for _, why in o.pre_step_hooks[0].__self__.last_launch_log] for k, o in opts2.items()}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"after2": sums, "after3": sums2, "logged": len([l for l in lines if l.startswith("Step: ")]),
                   "step": float(opts2["G"].step_t), "launch_logs": logs}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
