"""CPU, world_size 8 over gloo: `fed.federated_epoch` for the eight clients of configs/websetting/config_8gpu.csv —
BASELINE configs[2] on one 8 x MI355X node, one client per rank — with uneven shards (stragglers that run out of
batches, mid-epoch averages adopted only by clients that still train), weighted and unweighted averaging, and secure
aggregation under `PairwiseMasks` at K = 8 (28 pair keys from the X25519 agreement).  Every rank's final arena is
compared with a single-process replay of secure_aggregation_epoch's loop (torchlib/utils.py:1108-1233) on the oracle's
aggregation.  HIP arithmetic is replaced by the CPU stand-ins of tests/cpu_standins.py: what is under test is the
collective choreography that runs unchanged on RCCL."""
import os
import sys
from collections import OrderedDict
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SHARDS = [7, 5, 7, 2, 6, 1, 4, 7]        # batches per client: stragglers at 1, 2, 4, 5, 6


def make_args(weighted, secure):
    return SimpleNamespace(optimizer="SGD", lr=0.05, weight_decay=1e-3, sync_every_n_batch=3, keep_optim_dict=False,
                           weighted_averaging=weighted, unencrypted_aggregation=not secure, precision_fractional=16)


def client_batches(k, n_batches, dim):
    g = torch.Generator().manual_seed(1000 + k)
    return [(torch.randn(8, dim, generator=g), torch.randn(8, generator=g)) for _ in range(n_batches)]


def replay(args, world, dim, n_words):
    """secure_aggregation_epoch on one process: the reference's loop with the oracle's aggregation."""
    sys.path.insert(0, ROOT)
    from oracle import train_oracle as O
    from tests.cpu_standins import ToyEngine

    engines = [ToyEngine(n_words, dim, seed=0) for _ in range(world)]       # all start from the same model
    data = [client_batches(k, SHARDS[k], dim) for k in range(world)]
    total = sum(SHARDS)
    w = [SHARDS[k] / total for k in range(world)] if args.weighted_averaging else None
    losses = []

    def aggregate():
        sds = [OrderedDict(a=e.flat.clone()) for e in engines]
        if args.unencrypted_aggregation:
            return O.fedavg_plain(sds, w)["a"]
        return O.fedavg_secure(sds, w, args.precision_fractional)["a"]

    for b in range(max(SHARDS)):
        for k in range(world):
            if b < SHARDS[k]:
                x, y = data[k][b]
                engines[k].forward(x)
                losses.append(engines[k].loss_backward(y))
                engines[k].sgd_step(args.lr, args.weight_decay)
        if b > 0 and b % args.sync_every_n_batch == 0:
            avg = aggregate()
            for k in range(world):
                if SHARDS[k] > b:
                    engines[k].flat.copy_(avg)
    avg = aggregate()
    return avg, float(torch.stack(losses).double().mean())


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from primia_amd import fed
    from primia_amd.torchlib_compat import read_websocket_config
    from tests.cpu_standins import CpuArenaOps, CpuMaskOps, ToyEngine

    names = [w["id"] for w in read_websocket_config(os.path.join(ROOT, "configs", "websetting", "config_8gpu.csv")).values()]
    assert names[-1] == "crypto_provider" and len(names) - 1 == world
    dim, n_words = 16, 257
    res = {}
    for weighted in (False, True):
        for secure in (False, True):
            args = make_args(weighted, secure)
            eng = ToyEngine(n_words, dim, seed=0)
            masks = fed.PairwiseMasks.setup(n_words, "cpu", ops=CpuMaskOps()) if secure else None
            if masks is not None:
                res[f"pairs_{weighted}"] = len(masks.keys) == world - 1
            loader = client_batches(rank, SHARDS[rank], dim)
            loss, steps, local_flat, _ = fed.federated_epoch(eng, loader, args, ops=CpuArenaOps(), masks=masks)
            want, want_loss = replay(args, world, dim, n_words)
            tag = f"w{int(weighted)}s{int(secure)}"
            if secure:   # integer ring sum: order independent, bit exact
                res[tag] = torch.equal(local_flat, want) and torch.equal(eng.flat, want)
            else:        # float all-reduce: gloo's summation order is not the replay's
                res[tag] = torch.allclose(local_flat, want, rtol=1e-5, atol=1e-7) and torch.equal(eng.flat, local_flat)
            res[tag + "_steps"] = steps == SHARDS[rank]
            res[tag + "_loss"] = abs(loss - want_loss) < 1e-6 * max(1.0, abs(want_loss))
            res[tag + "_bn_counter"] = eng.num_batches_tracked["bn1"] == 0     # the adopted state dict carries 0
    # the masks hide an update: what one client contributes differs from its encoded arena, yet the sum is untouched
    masks = fed.PairwiseMasks.setup(64, "cpu", ops=CpuMaskOps())
    q = torch.arange(64, dtype=torch.int64) * (rank + 1)
    plain = q.clone()
    masks.apply(q)
    res["masked_differs"] = not torch.equal(q, plain)
    dist.all_reduce(q)
    res["masks_cancel"] = torch.equal(q, torch.arange(64, dtype=torch.int64) * sum(range(1, world + 1)))
    torch.save(res, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_federated_epoch_eight_clients(tmp_path):
    from tests.conftest import free_port

    mp.spawn(_worker, args=(8, free_port(), str(tmp_path)), nprocs=8, join=True)
    for r in range(8):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert all(res.values()), (r, {k: v for k, v in res.items() if not v})


def test_cpu_chacha_standin_matches_rfc8439():
    """The CPU stand-in of the keystream kernel reproduces RFC 8439 2.3.2 (same state layout as csrc/chacha.hip)."""
    import struct

    import numpy as np

    from tests.cpu_standins import chacha20_words

    k64 = struct.unpack("<4Q", bytes(range(32)))
    words = chacha20_words(k64 + (0x4a000000,), 1 | (0x09000000 << 32), 8)
    assert words.astype(np.uint64).tobytes()[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4"


def test_x25519_rfc7748_vectors():
    from primia_amd.fed import PairwiseMasks as P

    k = bytes.fromhex("a546e36bf0527c9d3b16154b82465edd62144c0ac1fc5a18506a2244ba449ac4")
    u = bytes.fromhex("e6db6867583030db3594c1a424b15f7c726624ec26b3353b10a903a6d0ab1c4c")
    assert P.x25519(k, u).hex() == "c3da55379de9c6908e94ea4df28d084f32eccf03491c71f754b4075577a28552"
    a = bytes.fromhex("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a")
    b = bytes.fromhex("5dab087e624a8a4b79e17f8b83800ee66f3bb1292618b6fd1c2f8b27ff88e0eb")
    nine = (9).to_bytes(32, "little")
    shared = "4a5d9d5ba4ce2de1728e3bf480350f25e07e21c947d19e3376f09b3c1e161742"
    assert P.x25519(a, P.x25519(b, nine)).hex() == shared and P.x25519(b, P.x25519(a, nine)).hex() == shared
    assert P.pair_key(bytes.fromhex(shared), 0, 1) != P.pair_key(bytes.fromhex(shared), 0, 2)
