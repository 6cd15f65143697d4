"""CPU, world_size 2 over gloo: the FedAvg choreography of primia_amd.fed (who contributes, who
adopts, weights, sync schedule, secure int64 path) against the oracle's restatement of
aggregation() (torchlib/utils.py:1000-1092).

The arena arithmetic of the product is HIP-only; here a torch-CPU stand-in for those four
element-wise ops is injected (test infrastructure), so what is under test is the collective
logic that runs unchanged on RCCL."""
import os
import sys
from collections import OrderedDict

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class CpuArenaOps:
    """Test-side stand-in for primia_amd.fed.HipArenaOps (same semantics as the kernels)."""

    def scale(self, x, a):
        x.mul_(torch.tensor(a, dtype=torch.float32))

    def divide(self, x, d):
        x.div_(torch.tensor(d, dtype=torch.float32))

    def encode(self, x, q, scale):
        q.copy_((x * torch.tensor(scale, dtype=torch.float32)).long())

    def decode(self, q, x, scale):
        x.copy_(q.float() / torch.tensor(scale, dtype=torch.float32))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import train_oracle as O
    from primia_amd import fed

    torch.manual_seed(100)  # same on both ranks: both can rebuild every client's arena
    arenas = [torch.randn(1003) * 0.1 for _ in range(world)]
    sds = [OrderedDict(a=a.clone()) for a in arenas]
    ops = CpuArenaOps()
    out = torch.empty(1003)
    res = {}
    # unweighted plaintext
    fed.fedavg_allreduce(arenas[rank], out, None, False, ops=ops)
    res["plain"] = torch.allclose(out, O.fedavg_plain(sds)["a"], rtol=1e-6, atol=1e-8)
    # weighted plaintext
    w = [0.25, 0.75]
    fed.fedavg_allreduce(arenas[rank], out, w[rank], False, ops=ops)
    res["weighted"] = torch.allclose(out, O.fedavg_plain(sds, w)["a"], rtol=1e-6, atol=1e-8)
    # secure: integer ring sum is order independent -> bit exact, at both precisions
    for pf in (16, 3):
        fed.fedavg_allreduce(arenas[rank], out, None, True, pf, 10, ops=ops)
        res[f"secure{pf}"] = torch.equal(out, O.fedavg_secure(sds, None, pf)["a"])
        fed.fedavg_allreduce(arenas[rank], out, w[rank], True, pf, 10, ops=ops)
        res[f"secure_w{pf}"] = torch.equal(out, O.fedavg_secure(sds, w, pf)["a"])
    res["input_untouched"] = torch.equal(arenas[rank], sds[rank]["a"])
    counts = fed.all_gather_int(3 + rank)
    res["gather"] = counts == [3, 4]
    # secure mean/std exchange of setup_pysyft (utils.py:764-794), fix_precision() = 10^3
    from oracle import datapipe_oracle as D
    means = [torch.tensor([0.48, 0.52, 0.4371]), torch.tensor([0.5125, 0.4999, 0.61])]
    stds = [torch.tensor([0.2219, 0.25, 0.3]), torch.tensor([0.19, 0.2451, 0.2777])]
    m, sd = fed.exchange_mean_std(means[rank], stds[rank], ops=ops)
    om, osd = D.exchange_mean_std(means, stds)
    res["mean_std_exchange"] = torch.equal(m, om) and torch.equal(sd, osd)
    # ... and against the values the reference's own tensor classes produced (tests/golden/mean_std_ref.npz)
    import numpy as np
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mean_std_ref.npz"))
    m, sd = fed.exchange_mean_std(torch.from_numpy(gold["w2c1/means"][rank]), torch.from_numpy(gold["w2c1/stds"][rank]),
                                  ops=ops)
    res["mean_std_reference"] = (np.array_equal(m.numpy(), gold["w2c1/mean"])
                                 and np.array_equal(sd.numpy(), gold["w2c1/std"]))
    torch.save(res, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_fedavg_allreduce_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert all(res.values()), (r, res)


def test_sync_schedule_matches_reference_loop():
    """secure_aggregation_epoch (utils.py:1159-1230): sync when batch_idx > 0 and batch_idx % n == 0;
    mid-epoch broadcasts only reach workers with num_batches > batch_idx; exhausted workers skip."""
    from primia_amd.fed import SyncSchedule

    s = SyncSchedule([7, 4, 2], 3)
    assert s.max_batches == 7
    assert [b for b in range(7) if s.sync_after(b)] == [3, 6]
    assert [s.trains(1, b) for b in range(7)] == [True] * 4 + [False] * 3
    assert [c for c in range(3) if s.adopts(c, 3)] == [0, 1]
    assert [c for c in range(3) if s.adopts(c, 6)] == [0]
    s1 = SyncSchedule([5], 1)
    assert [b for b in range(5) if s1.sync_after(b)] == [1, 2, 3, 4]
