"""One rank of the per-rank federated tests (launched through torch.distributed.run by test_gpu_federated.py; the ranks
share GPU 0 and talk over gloo).  Trains this rank's client with fed.federated_epoch and writes its final arena."""
import json
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primia_amd import fed, resnet_spec as rs  # noqa: E402
from primia_amd.engine import ResNet18Engine  # noqa: E402


def shard(rank, n_batches, batch, size):
    g = torch.Generator().manual_seed(100 + rank)
    return [(torch.randn(batch, 3, size, size, generator=g), torch.randint(0, 3, (batch,), generator=g))
            for _ in range(n_batches)]


if __name__ == "__main__":
    cfg = json.loads(sys.argv[1])
    out = sys.argv[2]
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    args = SimpleNamespace(**cfg["args"])
    batch, size = cfg["batch"], cfg["size"]
    torch.manual_seed(11)
    init = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"))
    norm = cfg.get("norm", "batch")
    init = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"), norm) if norm != "batch" else init
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=device, norm=norm)
    if cfg.get("dp"):
        eng.dp_params = dict(cfg["dp"])
    eng.load_state_dict(init)
    loader = [(x.to(device), y.to(device)) for x, y in shard(rank, cfg["shards"][rank], batch, size)]
    masks = None
    if not args.unencrypted_aggregation and cfg.get("masks", True):
        masks = fed.PairwiseMasks.setup(eng.flat.numel(), device)
    opt, local_flat, losses = None, None, []
    for epoch in range(cfg["epochs"]):
        loss, steps, local_flat, opt = fed.federated_epoch(eng, loader, args, opt, local_flat=local_flat, masks=masks)
        losses.append(loss)
    if masks is not None:
        # what this rank put on the wire at the last sync differs from its plain encoding, and the masks cancel
        n = 4096
        q = torch.arange(n, dtype=torch.int64, device=device)
        before = q.clone()
        masks.apply(q)
        assert not torch.equal(q, before)
        dist.all_reduce(q)
        assert torch.equal(q, before * dist.get_world_size())
    torch.save({"flat": eng.flat.cpu(), "local": local_flat.cpu(), "losses": losses,
                "opt": opt.state_dict(), "steps": eng.opt_steps, "nbt": dict(eng.num_batches_tracked)}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()
