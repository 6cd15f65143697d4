"""GPU: the in-process federated epoch (reference layout: all clients in one process) and the
PriMIA-compatible CLIs."""
import copy
import json
import os
import subprocess
import sys
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

from oracle import train_oracle as O  # noqa: E402
from primia_amd import resnet_spec as rs  # noqa: E402
from primia_amd.engine import ResNet18Engine  # noqa: E402
from primia_amd.torchlib_compat import aggregation, send_new_models, train_federated  # noqa: E402
from tests.conftest import free_port  # noqa: E402


def make_args(**kw):
    base = dict(optimizer="SGD", weight_decay=5e-4, sync_every_n_batch=1, keep_optim_dict=False,
                weighted_averaging=False, unencrypted_aggregation=True, precision_fractional=16, lr=1e-3)
    base.update(kw)
    return SimpleNamespace(**base)


def flat_of(sd, eng):
    return torch.cat([sd[k].flatten().float() for k, _ in eng.p_entries + eng.b_entries])


@pytest.mark.parametrize("weighted", [False, True])
def test_aggregation_matches_oracle(cuda, weighted):
    """aggregation() on arenas == the oracle's per-key restatement: plaintext to 1e-6, secure
    (encode -> ring sum -> decode) BIT-EXACT at precision 16 and 3."""
    torch.manual_seed(3)
    engs = {}
    for w in ("alice", "bob", "charlie"):
        e = ResNet18Engine(2, 3, 3, 64, "max", dtype=torch.float32, device=cuda)
        e.init_weights()
        engs[w] = e
    local = ResNet18Engine(2, 3, 3, 64, "max", dtype=torch.float32, device=cuda)
    sds = [engs[w].state_dict() for w in engs]
    weights = {"alice": 0.5, "bob": 0.3, "charlie": 0.2} if weighted else None
    wl = [weights[w] for w in engs] if weighted else None
    models = dict(engs)
    aggregation(local, models, list(engs), None, make_args(), None, weights=weights, secure=False)
    want = O.fedavg_plain(sds, wl)
    got = local.state_dict()
    for k in want:
        assert torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-8), k
    for pf in (16, 3):
        aggregation(local, models, list(engs), None, make_args(precision_fractional=pf), None, weights=weights,
                    secure=True)
        want = O.fedavg_secure(sds, wl, pf)
        got = local.state_dict()
        for k in want:
            assert torch.equal(got[k], want[k]), (pf, k)
    send_new_models(local, models)
    assert all(torch.equal(e.flat, local.flat) for e in engs.values())


def test_federated_epoch_tracks_oracle(cuda):
    """Two clients with uneven shards, sync every batch, plaintext FedAvg: the averaged model after
    one epoch follows an oracle simulation of secure_aggregation_epoch on CPU."""
    size, batch, lr, wd = 64, 4, 1e-3, 5e-4
    torch.manual_seed(11)
    init = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"))
    g = torch.Generator().manual_seed(12)
    shards = {"alice": 3, "bob": 2}
    data = {w: [(torch.randn(batch, 3, size, size, generator=g), torch.randint(0, 3, (batch,), generator=g))
                for _ in range(n)] for w, n in shards.items()}
    args = make_args(lr=lr, weight_decay=wd, sync_every_n_batch=1)
    models = {"local_model": ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda)}
    models["local_model"].load_state_dict(init)
    for w in shards:
        models[w] = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda)
        models[w].load_state_dict(init)
    loaders = {w: [(x.to(cuda), y.to(cuda)) for x, y in data[w]] for w in shards}
    opt = {w: {"lr": lr} for w in shards}
    models = train_federated(args, models, cuda, loaders, opt, 1, {w: None for w in shards}, None, verbose=False)
    got = models["local_model"].state_dict()
    # ---- oracle simulation of the same loop -------------------------------------------------------
    sds = {w: OrderedDict((k, v.clone()) for k, v in init.items()) for w in shards}
    nb = {w: len(d) for w, d in data.items()}

    def agg():
        avg = O.fedavg_plain([sds[w] for w in shards])
        return avg

    for b in range(max(nb.values())):
        for w in shards:
            if b < nb[w]:
                O.train_step(sds[w], data[w][b][0], data[w][b][1], lr, wd)
        if b > 0 and b % 1 == 0:
            avg = agg()
            for w in shards:
                if nb[w] > b:
                    for k in avg:
                        sds[w][k] = avg[k].clone()
    avg = agg()
    # weights agree to 1e-5 of their norm plus 5% of the epoch's UPDATE: end-to-end gradients carry the
    # ReLU-flip discontinuity documented in test_gpu_train_step.py (here BN sees only 16-64 samples
    # per channel at batch 4), and the update is lr * sum of those gradients.
    for k in avg:
        a, r, i0 = got[k].double(), avg[k].double(), init[k].double()
        assert (a - r).norm() <= 1e-5 * r.norm() + 0.05 * (r - i0).norm() + 1e-7, k
    # both clients ended on the average
    assert torch.equal(models["alice"].flat, models["local_model"].flat)


def run(cmd, env=None):
    e = dict(os.environ, PRIMIA_ALLOW_RANDOM_INIT="1")   # the smoke preset says pretrained = yes; no ImageNet file here
    e.update(env or {})
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_cli_train_federated_then_inference(tmp_path):
    env = {"PRIMIA_SYNTHETIC_BATCHES": "3", "PRIMIA_DTYPE": "bf16"}
    out = run(["train.py", "--config", "configs/torch/smoke-federated.ini", "--train_federated", "--cuda",
               "--training_name", "clitest", "--data_dir", "synthetic"], env)
    assert "Train Epoch: 2" in out and "matthews coeff" in out   # the reference's validation table
    ckpt = os.path.join(ROOT, "model_weights", "final_federated_clitest.pt")
    assert os.path.exists(ckpt)
    state = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert set(state) == {"epoch", "model_state_dict", "optim_state_dict", "args", "val_mean_std"}
    assert len(state["model_state_dict"]) == 122
    for flags in ([], ["--encrypted_inference"]):
        out = run(["inference.py", "--model_weights", ckpt, "--data_dir", "synthetic", "--num_images", "2", "--cuda"]
                  + flags)
        res = json.loads(out.strip().splitlines()[-1])["Inference Results"]
        assert sorted(res) == ["0", "1"] and all(v in (0, 1, 2) for v in res.values())
    # the same encrypted inference with the three roles as three ranks (all on GPU 0 over gloo here): logits
    # bit-identical to the in-process run's under the same (debug) dealer seed
    local, dist3 = str(tmp_path / "local.pt"), str(tmp_path / "dist.pt")
    base = ["--model_weights", ckpt, "--data_dir", "synthetic", "--num_images", "2", "--cuda", "--encrypted_inference",
            "--debug_dealer_seed", "0"]
    run(["inference.py"] + base, {"PRIMIA_DUMP_LOGITS": local})
    out = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "inference.py"] + base + ["--three_role"],
              {"PRIMIA_DUMP_LOGITS": dist3, "MASTER_ADDR": "127.0.0.1"})
    assert "Inference Results" in out
    assert torch.equal(torch.load(local), torch.load(dist3))
    # ... and as the serving form (--hip_graph: captured online phase + one-launch dealer refill per image): other primitives,
    # so the logits agree up to fixed-point noise — for EVERY image (the graphed form returns a static buffer; the CLI keeps
    # copies: round 6 found the dump holding the last image's logits twice)
    graphed = str(tmp_path / "graph.pt")
    out = run(["inference.py"] + base + ["--hip_graph"], {"PRIMIA_DUMP_LOGITS": graphed})
    lg, ll = torch.load(graphed), torch.load(local)
    assert lg.shape == ll.shape == (2, 3)
    assert torch.allclose(lg, ll, atol=5e-3), (lg, ll)
    print("logits, eager:", ll.tolist(), "graphed:", lg.tolist())
    if not torch.equal(ll[0], ll[1]):           # (a barely trained model may not tell two noise images apart at all)
        assert not torch.equal(lg[0], lg[1])
    os.remove(ckpt)


@pytest.mark.parametrize("secure", [False, True])
def test_bench_two_ranks_control_flow(tmp_path, secure):
    """bench.py under torch.distributed.run with 2 ranks.  This box has ONE GPU, so both ranks share it
    and talk over gloo; what is exercised is the N > 1 path of the benchmark (barriers, FedAvg every 3
    steps, max-over-ranks timing, rank-0 JSON) that the driver runs on RCCL across 8 GPUs."""
    env = dict(os.environ, PRIMIA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--steps", "7",
           "--warmup", "1", "--batch", "8", "--size", "64", "--no-cpu-baseline"] + (["--secure-aggregation"] if secure else [])
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["clients"] == 2
    assert d["config"]["secure_aggregation"] is secure
    assert d["hip_graph"] is True   # the local step is replayed as a graph at every rank count, the exchange is not
    assert d["value"] > 0 and abs(d["value"] - 2 * d["images_per_sec_per_client"]) < 1e-3 * d["value"]


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent (which never touches the GPU) starts
    the two ranks as a child under torch.distributed.run, relays rank 0's JSON line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PRIMIA_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "8", "--size", "64",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines          # ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["clients"] == 2 and d["fedavg_ms_per_sync"] is not None
    # the N > 1 line proves which collective ran on how many ranks (RCCL on a multi-GPU node; gloo on this one-GPU box)
    # and carries BASELINE.md B2 beside it
    assert d["collective"]["ranks"] == 2 and d["collective"]["backend"] == "gloo" and d["collective"]["gpus_seen"]
    assert d["cpu_baseline_b2"]["K"] == 2
    assert d["roofline"]["kernel"] in d["roofline"]["kernels"] and "layers_us" in d["roofline"]


def test_aggregation_kernels_match_reference_aggregation(cuda, golden_dir):
    """torchlib_compat.aggregation (scale / fx_encode / ring_add / fx_decode / divide kernels on a flat arena) against
    the outputs of the reference's own aggregation() (tests/golden/fedavg_ref.npz).  Secure path: bit-exact; plain
    path: 1e-6 (the reference sums a stacked tensor, the arena accumulates client by client)."""
    import numpy as np
    from types import SimpleNamespace

    from primia_amd.torchlib_compat import aggregation

    z = np.load(os.path.join(golden_dir, "fedavg_ref.npz"))
    keys = [k.split("/", 1)[1] for k in z.files if k.startswith("in0/") and not k.endswith("num_batches_tracked")]
    ids = ["alice", "bob", "charlie"]

    class Arena:
        def __init__(self, flat):
            self.flat, self.num_batches_tracked = flat, {"bn": 7}

        def refresh_weights(self):
            pass

    def flat_of(c):
        return torch.cat([torch.from_numpy(z[f"in{c}/{k}"]).reshape(-1) for k in keys]).to(cuda)

    w = {"alice": 0.2, "bob": 0.3, "charlie": 0.5}
    for tag, secure, weights, pf in (("plain.u.p0", False, None, 16), ("plain.w.p0", False, w, 16),
                                     ("secure.u.p3", True, None, 3), ("secure.u.p16", True, None, 16),
                                     ("secure.w.p3", True, w, 3), ("secure.w.p16", True, w, 16)):
        models = {i: Arena(flat_of(c)) for c, i in enumerate(ids)}
        local = Arena(torch.zeros_like(models["alice"].flat))
        aggregation(local, models, ids, None, SimpleNamespace(precision_fractional=pf), None, weights=weights,
                    secure=secure)
        want = torch.cat([torch.from_numpy(z[f"{tag}/{k}"]).reshape(-1) for k in keys])
        got = local.flat.cpu()
        if secure:
            assert torch.equal(got, want), tag
        else:
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-7), tag
        assert local.num_batches_tracked == {"bn": 0}


def _in_process_epochs(cuda, cfg):
    """The same clients, shards and settings as tests/fed_rank_worker.py, all in this process
    (torchlib_compat.train_federated, the reference's VirtualWorker layout)."""
    from tests.fed_rank_worker import shard

    args = SimpleNamespace(**cfg["args"])
    batch, size = cfg["batch"], cfg["size"]
    torch.manual_seed(11)
    norm = cfg.get("norm", "batch")
    init = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"))
    if norm != "batch":
        init = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"), norm)

    def make():
        e = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda, norm=norm)
        if cfg.get("dp"):
            e.dp_params = dict(cfg["dp"])
        e.load_state_dict(init)
        return e

    names = [f"w{k}" for k in range(len(cfg["shards"]))]
    models = {"local_model": make()}
    loaders, opt = {}, {}
    for k, w in enumerate(names):
        models[w] = make()
        loaders[w] = [(x.to(cuda), y.to(cuda)) for x, y in shard(k, cfg["shards"][k], batch, size)]
        opt[w] = {"lr": args.lr}
    for epoch in range(cfg["epochs"]):
        models = train_federated(args, models, cuda, loaders, opt, epoch + 1, {w: None for w in names}, None,
                                 verbose=False)
    return models, opt, names


def test_mean_std_exchange_matches_reference_tensors(cuda, golden_dir):
    """8f.1: fed.secure_mean_of (HIP encode -> ring add -> decode -> / K) against the values the reference's own
    FixedPrecisionTensor / AdditiveSharingTensor classes produced for utils.py:764-794 (mean_std_ref.npz)."""
    from primia_amd import fed

    gold = np.load(os.path.join(golden_dir, "mean_std_ref.npz"))
    for tag in ("w2c1", "w3c3", "w5c3"):
        stats = [(torch.from_numpy(m).to(cuda), torch.from_numpy(s).to(cuda))
                 for m, s in zip(gold[tag + "/means"], gold[tag + "/stds"])]
        m, s = fed.secure_mean_of(stats)
        assert np.array_equal(m.cpu().numpy(), gold[tag + "/mean"]), tag
        assert np.array_equal(s.cpu().numpy(), gold[tag + "/std"]), tag


@pytest.mark.parametrize("case", ["secure_sgd", "plain_adam_keep", "weighted_secure_adam_reset", "dp_clip_secure"])
def test_per_rank_federated_epoch_matches_in_process(cuda, tmp_path, case):
    """one client per rank (here 2 ranks sharing GPU 0 over gloo; RCCL on a multi-GPU node) against the
    in-process federated epoch on the same uneven shards — stragglers (3 vs 2 batches, a sync after every batch),
    optimizer re-creation vs keep_optim_dict, weighted averaging, pairwise-masked secure aggregation.
    Secure aggregation is integer arithmetic: the final arenas are BIT-identical.  Plain unweighted averaging of two
    clients is too (a + b is commutative); weighted plain differs by the rounding of scale-then-add vs fused axpy."""
    a = dict(optimizer="SGD", lr=1e-3, weight_decay=5e-4, sync_every_n_batch=1, keep_optim_dict=False,
             weighted_averaging=False, unencrypted_aggregation=False, precision_fractional=16, beta1=0.5, beta2=0.99)
    if case == "plain_adam_keep":
        a.update(optimizer="Adam", keep_optim_dict=True, unencrypted_aggregation=True)
    elif case == "weighted_secure_adam_reset":
        a.update(optimizer="Adam", weighted_averaging=True, sync_every_n_batch=2)
    cfg = {"args": a, "batch": 4, "size": 64, "shards": [3, 2], "epochs": 2}
    if case == "dp_clip_secure":
        # BASELINE configs[3]'s control path: GroupNorm network, per-sample clipping (noise off: each deployment would
        # draw its own), secure aggregation — clients of a DP run are engines like any other to the federated epoch
        cfg.update(norm="group", dp={"max_grad_norm": 1.0, "noise_multiplier": 0.0})
    out = str(tmp_path / "rank")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "fed_rank_worker.py"),
           json.dumps(cfg), out]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    models, opt, names = _in_process_epochs(cuda, cfg)
    want = models["local_model"].flat.cpu()
    for k, w in enumerate(names):
        got = torch.load(f"{out}.{k}", weights_only=False)
        assert torch.equal(got["flat"], got["local"])            # every client ends the epoch on the average
        assert torch.equal(got["flat"], want), (case, k, (got["flat"] - want).abs().max().item())
        assert got["steps"] == models[w].opt_steps                # optimizer re-creation / keep_optim_dict
        assert got["nbt"] == models[w].num_batches_tracked
        mine = opt[w].state_dict()
        assert got["opt"]["param_groups"] == mine["param_groups"]
        assert set(got["opt"]["state"]) == set(mine["state"])
        for i in mine["state"]:
            assert torch.equal(got["opt"]["state"][i]["exp_avg"], mine["state"][i]["exp_avg"])


def test_optimizer_state_dict_round_trips_with_torch_adam(cuda):
    """8f.3: a checkpoint written with a stock torch.optim.Adam.state_dict() (index-keyed as torch >= 1.5 writes it, or
    id()-keyed as the reference's torch 1.4 does) restores the engine's Adam moments: the next step equals torch's."""
    from primia_amd.optim import EngineOptimizer

    size, batch = 64, 2
    torch.manual_seed(5)
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda)
    eng.init_weights()
    params = [torch.nn.Parameter(eng.views[k].detach().cpu().clone()) for k, _ in eng.p_entries]
    topt = torch.optim.Adam(params, lr=1e-3, betas=(0.5, 0.99), weight_decay=5e-4)
    g = torch.Generator().manual_seed(6)
    grads = [[torch.randn(p.shape, generator=g) * 0.01 for p in params] for _ in range(3)]
    for step in range(2):
        for p, gr in zip(params, grads[step]):
            p.grad = gr.clone()
        topt.step()
    sd = topt.state_dict()
    for keyed in ("index", "id"):
        saved = {"state": dict(sd["state"]), "param_groups": [dict(sd["param_groups"][0])]}
        if keyed == "id":       # torch 1.4: keys are id(param)
            remap = {i: 140000000000000 + 64 * i for i in saved["param_groups"][0]["params"]}
            saved["state"] = {remap[i]: v for i, v in saved["state"].items()}
            saved["param_groups"][0]["params"] = [remap[i] for i in saved["param_groups"][0]["params"]]
        for k, _ in eng.p_entries:
            eng.views[k].copy_(params[[n for n, _ in eng.p_entries].index(k)].detach().to(cuda))
        opt = EngineOptimizer(eng, "Adam", lr=9.9, betas=(0.9, 0.999))
        opt.load_state_dict(saved)
        assert opt.param_groups[0]["lr"] == 1e-3 and tuple(opt.param_groups[0]["betas"]) == (0.5, 0.99)
        assert eng.opt_steps == 2
        again = opt.state_dict()
        assert sorted(again["state"]) == list(range(len(params)))
        assert torch.equal(again["state"][3]["exp_avg"], sd["state"][3]["exp_avg"])
        for (k, _), gr in zip(eng.p_entries, grads[2]):
            eng.gviews[k].copy_(gr.to(cuda))
        opt.step()
        ref = [p.detach().clone() for p in params]
        ropt = torch.optim.Adam([torch.nn.Parameter(r) for r in ref], lr=1e-3, betas=(0.5, 0.99), weight_decay=5e-4)
        ropt.load_state_dict(copy.deepcopy(sd))      # torch's loader keeps the `step` tensors it is given
        for p, gr in zip(ropt.param_groups[0]["params"], grads[2]):
            p.grad = gr.clone()
        ropt.step()
        for (k, _), p in zip(eng.p_entries, ropt.param_groups[0]["params"]):
            a, b = eng.views[k].cpu().double().flatten(), p.detach().double().flatten()
            assert (a - b).norm() <= 1e-5 * b.norm() + 1e-9, k
    # SGD: no per-parameter state; a checkpoint that carries some is refused loudly
    sgd = EngineOptimizer(eng, "SGD", lr=1e-2, weight_decay=5e-4)
    assert sgd.state_dict()["state"] == {}
    with pytest.raises(ValueError):
        sgd.load_state_dict(sd)


def _write_tree(root, workers=2, per_class=4):
    from tests.test_gpu_imagefolder import make_folder

    for i in range(workers):
        make_folder(os.path.join(root, f"worker{i + 1}"), per_class=per_class, seed=10 + i)
    make_folder(os.path.join(root, "validation"), per_class=3, seed=99)


def test_cli_per_rank_training_on_a_real_folder_and_resume(tmp_path):
    """train.py under torch.distributed.run: rank k = client k of the CSV on its own process (2 ranks sharing GPU 0 over
    gloo here), real image folders, secure aggregation under pairwise masks, Adam; then the resume matrix of
    train.py:344-389: federated -> federated (per-worker optimizer state restored) and federated -> vanilla."""
    data = str(tmp_path / "data")
    _write_tree(data)
    csv = tmp_path / "workers.csv"
    csv.write_text("id,alice,bob,crypto_provider\nhost,127.0.0.1,127.0.0.1,127.0.0.1\nport,8777,8778,8780\n")
    ini = tmp_path / "cfg.ini"
    text = open(os.path.join(ROOT, "configs", "torch", "smoke-federated.ini")).read()
    text = (text.replace("epochs = 10", "epochs = 2").replace("repetitions_dataset = 5", "repetitions_dataset = 1")
            .replace("optimizer = SGD", "optimizer = Adam").replace("sync_every_n_batch = 3", "sync_every_n_batch = 1")
            .replace("keep_optim_dict = no", "keep_optim_dict = yes"))
    ini.write_text(text)
    env = {"PRIMIA_BACKEND": "gloo", "PRIMIA_WEBSOCKETS_CONFIG": str(csv), "MASTER_ADDR": "127.0.0.1"}
    torchrun = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(free_port())]
    base = ["train.py", "--config", str(ini), "--cuda", "--data_dir", data]
    out = run(torchrun + base + ["--train_federated", "--training_name", "rankcli"], env)
    assert "Train Epoch: 2" in out and "matthews coeff" in out
    assert out.count("Train Epoch: 1") == 1                     # only rank 0 reports
    ckpt = os.path.join(ROOT, "model_weights", "final_federated_rankcli.pt")
    state = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert set(state["optim_state_dict"]) == {"alice", "bob"}
    for w in ("alice", "bob"):
        o = state["optim_state_dict"][w]
        assert set(o) == {"state", "param_groups"} and len(o["state"]) == 62 and o["state"][0]["step"] > 0
        assert o["param_groups"][0]["betas"] == (0.5, 0.99)
    m, s = state["val_mean_std"]
    assert m.shape == (3,) and 0.3 < float(m.mean()) < 0.7 and 0.1 < float(s.mean()) < 0.5   # real statistics
    assert state["args"].train_federated and len(state["model_state_dict"]) == 122
    # federated -> federated, in process this time: starts AT the saved epoch with the saved optimizer state
    out = run(base + ["--train_federated", "--training_name", "resumed", "--resume_checkpoint", ckpt],
              {"PRIMIA_WEBSOCKETS_CONFIG": str(csv)})
    assert "Resume training from a given checkpoint." in out
    # (which epoch the "final" checkpoint carries is the best validation epoch: data- and rounding-dependent, 1 or 2 here)
    assert "Train Epoch: {:d}".format(state["epoch"]) in out
    assert all("Train Epoch: {:d} ".format(e) not in out for e in range(1, state["epoch"]))
    # federated -> vanilla
    out = run(base[:4] + ["--data_dir", os.path.join(data, "worker1"), "--training_name", "vanilla",
                          "--resume_checkpoint", ckpt])
    assert "Resume training" in out
    van = os.path.join(ROOT, "model_weights", "final_vanilla_vanilla.pt")
    vs = torch.load(van, map_location="cpu", weights_only=False)
    assert set(vs["optim_state_dict"]) == {"state", "param_groups"}
    for f in (ckpt, van, os.path.join(ROOT, "model_weights", "final_federated_resumed.pt")):
        os.remove(f)


def test_cli_runs_the_fast_preset_with_every_augmentation_on(tmp_path):
    """configs/torch/pneumonia-resnet-pretrained-fast.ini — the reference preset that switches EVERY albumentations member
    on — through train.py --train_federated on real image folders (two clients in one process, MixUp at registration as
    the reference does), with nothing edited but the sizes (64-pixel images, batch 8) and the per-transform probabilities
    raised so that every member fires within the few images of the test."""
    data = str(tmp_path / "data")
    _write_tree(data, workers=2, per_class=8)
    csv = tmp_path / "workers.csv"
    csv.write_text("id,alice,bob,crypto_provider\nhost,127.0.0.1,127.0.0.1,127.0.0.1\nport,8777,8778,8780\n")
    ini = tmp_path / "fast.ini"
    text = open(os.path.join(ROOT, "configs", "torch", "pneumonia-resnet-pretrained-fast.ini")).read()
    for a_, b_ in (("batch_size = 200", "batch_size = 8"), ("train_resolution = 224", "train_resolution = 64"),
                   ("overall_prob = 0.75", "overall_prob = 1.0"), ("individual_probs = 0.2", "individual_probs = 0.6")):
        assert a_ in text, a_
        text = text.replace(a_, b_)
    ini.write_text(text)
    out = run(["train.py", "--config", str(ini), "--cuda", "--data_dir", data, "--train_federated", "--training_name",
               "fastpreset"], {"PRIMIA_WEBSOCKETS_CONFIG": str(csv)})
    assert "Train Epoch: 1" in out and "matthews coeff" in out and "not part of the accelerated data path" not in out
    os.remove(os.path.join(ROOT, "model_weights", "final_federated_fastpreset.pt"))


def test_cli_local_training_with_mixup_halving_batches(tmp_path):
    """train.py WITHOUT --train_federated on the shipped default preset's MixUp settings (mixup = yes, mixup_prob = 0.9:
    a batch arrives whole with probability 0.1 and as its mixed halves otherwise, torchlib/utils.py:1262-1267)."""
    data = str(tmp_path / "data")
    _write_tree(data, workers=1, per_class=12)
    ini = tmp_path / "local.ini"
    text = open(os.path.join(ROOT, "configs", "torch", "pneumonia-resnet-pretrained.ini")).read()
    for a_, b_ in (("batch_size = 200", "batch_size = 8"), ("train_resolution = 224", "train_resolution = 64"),
                   ("epochs = 40", "epochs = 3"), ("mixup_prob = 0.9", "mixup_prob = 0.5")):
        assert a_ in text, a_
        text = text.replace(a_, b_)
    ini.write_text(text)
    out = run(["train.py", "--config", str(ini), "--cuda", "--data_dir", os.path.join(data, "worker1"), "--training_name",
               "localmix"])
    assert "Train Epoch: 3" in out and "matthews coeff" in out
    os.remove(os.path.join(ROOT, "model_weights", "final_vanilla_localmix.pt"))


def test_c_abi_comm_entry_points_on_an_rccl_communicator(cuda):
    """§8b "comm": primia_fedavg_allreduce / primia_open2 take the CALLER's ncclComm_t.  A one-rank communicator made
    with the RCCL that torch ships (ctypes: ncclGetUniqueId / ncclCommInitRank) exercises the whole call path on one
    GPU: on one rank the sum is the rank's own buffer, so the results are exactly scale -> [encode -> decode] -> divide."""
    import ctypes

    from primia_amd import _lib
    from primia_amd._lib import call, query

    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid, comm = UniqueId(), ctypes.c_void_p()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    torch.cuda.set_device(cuda)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    assert query("primia_comm_available") == 1
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(10001, generator=g) * 3).to(cuda)
    for weight, secure, pf in ((-1.0, 0, 16), (0.25, 0, 16), (-1.0, 1, 16), (0.25, 1, 3)):
        flat = x.clone()
        scratch = torch.empty(flat.numel(), dtype=torch.int64, device=cuda)
        call("primia_fedavg_allreduce", flat, flat.numel(), weight, 1, secure, pf, scratch, comm)
        want = x.clone()
        if weight >= 0:
            call("primia_scale", want, want.numel(), weight)
        if secure:
            q = torch.empty(want.numel(), dtype=torch.int64, device=cuda)
            call("primia_fx_encode", want, q, want.numel(), float(10 ** pf))
            call("primia_fx_decode", q, want, want.numel(), float(10 ** pf))
        if weight < 0:
            call("primia_divide", want, want.numel(), 1.0)
        torch.cuda.synchronize()
        assert torch.equal(flat, want), (weight, secure, pf)
    buf = torch.arange(-500, 500, dtype=torch.int64, device=cuda) * (2 ** 53)
    keep = buf.clone()
    call("primia_open2", buf, buf.numel(), comm)
    torch.cuda.synchronize()
    assert torch.equal(buf, keep)
    rccl.ncclCommDestroy(comm)


def test_reference_style_setup_through_the_worker_shim(cuda):
    """A set-up written like the reference's (train.py:88-97, torchlib/utils.py:516-860) on primia_syft_compat: hook,
    setup_pysyft -> (train_loader keyed by VirtualWorker, val_loader, total_L, workers, names, crypto_provider,
    val_mean_std), then one federated epoch of torchlib_compat.train_federated on exactly those objects."""
    import primia_syft_compat as sy
    from primia_amd.optim import EngineOptimizer

    args = SimpleNamespace(batch_size=4, train_resolution=64, inference_resolution=64, seed=1, pretrained=True,
                           unencrypted_aggregation=False, websockets=False, data_dir="synthetic", mixup=False,
                           weight_classes=False, train_federated=True, repetitions_dataset=1, lr=1e-3, end_lr=1e-4,
                           optimizer="SGD", weight_decay=5e-4, beta1=0.9, beta2=0.99, sync_every_n_batch=2,
                           weighted_averaging=True, keep_optim_dict=False, precision_fractional=16, log_interval=10,
                           differentially_private=False)
    hook = sy.TorchHook(torch)
    train_loader, val_loader, total_L, workers, names, crypto_provider, val_mean_std = sy.setup_pysyft(
        args, hook, device=cuda, websockets_config=os.path.join(ROOT, "configs", "websetting", "config.csv"))
    assert names == ["alice", "bob", "charlie"] and crypto_provider.id == "crypto_provider"
    assert set(train_loader) == set(workers.values()) and total_L == 4 * (4 + 3 + 2)
    assert [len(train_loader[workers[n]]) for n in names] == [4, 3, 2]
    assert val_mean_std.shape == (2, 3) and torch.equal(val_mean_std[0], torch.zeros(3))
    torch.manual_seed(3)
    init = rs.init_state_dict(rs.resnet18_spec(3, 3, 64, "max"))

    def make():
        e = ResNet18Engine(4, 3, 3, 64, "max", dtype=torch.float32, device=cuda)
        e.load_state_dict(init)
        return e

    model = {"local_model": make(), **{w.id: make() for w in workers.values()}}
    optimizer = {w.id: EngineOptimizer.from_args(model[w.id], args) for w in workers.values()}
    loss_fn = {w.id: None for w in workers.values()}
    model = train_federated(args, model, cuda, train_loader, optimizer, 1, loss_fn, crypto_provider, verbose=False)
    new = model["local_model"].state_dict()
    assert any(not torch.equal(new[k], init[k]) for k in ("conv1.weight", "fc.weight"))
    for w in workers.values():          # every client adopted the last average
        assert torch.equal(model[w.id].flat, model["local_model"].flat)
