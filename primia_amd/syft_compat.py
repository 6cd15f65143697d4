"""The PySyft worker-facing objects PriMIA's federated set-up code touches, over device-resident tensors.

The reference builds its federation out of PySyft 0.2.x objects (torchlib/utils.py:516-860):

    hook = sy.TorchHook(torch)                                    train.py:88
    workers = {id: sy.VirtualWorker(hook, id=id, verbose=False)}  utils.py:578-581
    workers[w].object_store.clear_objects()                       utils.py:582-583
    mean.tag("#datamean"); worker.load_data([mean, std])          utils.py:691-693
    selected_data.tag("#traindata"); worker.load_data([...])      utils.py:735-741
    grid = sy.PrivateGridNetwork(*workers, crypto_provider)       utils.py:742-745
    data = grid.search("#traindata")                              utils.py:746-747   -> {worker id: [tensor]}
    sy.FederatedDataLoader(sy.FederatedDataset([sy.BaseDataset(data[w][0], target[w][0])]),
                           batch_size=args.batch_size, shuffle=True)                utils.py:750-760

In PySyft these are a message-passing control plane (serde, pointers, websocket or in-process transport; syft/workers/
base.py:192,229, syft/grid/private_grid.py:24) — out of scope here: on one node a federated client IS a
GPU, a "worker" is a named tensor store on that GPU, and `search` is a dictionary lookup.  What this module keeps is the
SHAPE of the API, so that a script written against the reference's set-up code builds the same objects and hands the
same `(train_loader, val_loader, total_L, workers, worker_names, crypto_provider, val_mean_std)` to the training loops of
primia_amd.torchlib_compat (which accept worker objects or their ids as keys).  No arithmetic lives here.

    import primia_syft_compat as sy          # instead of `import syft as sy`
"""
import torch

hook = None              # the last TorchHook built (PySyft: sy.hook)
local_worker = None      # ... and its local worker (sy.local_worker)

__all__ = ["TorchHook", "VirtualWorker", "ObjectStore", "PrivateGridNetwork", "BaseDataset", "FederatedDataset",
           "FederatedDataLoader", "setup_pysyft"]


class TorchHook:
    """sy.TorchHook(torch) (syft/frameworks/torch/hook/hook.py): PySyft monkey-patches every tensor method; the only
    additions the reference's set-up code uses are `tensor.tag(*tags)` / `tensor.tags`, installed here (idempotent)."""

    def __init__(self, torch_module=torch, local_worker=None, is_client=True, verbose=False):
        self.local_worker = local_worker or VirtualWorker(self, id="me")
        # PySyft publishes the hook and its local worker as module globals (`sy.hook`, `sy.local_worker`; the reference
        # reads `sy.hook` when torch is already hooked, train.py:84-87, and sets `sy.local_worker.clients` /
        # `.object_store.garbage_delay` in inference.py:158,233) and marks torch as hooked
        import sys

        for mod in (sys.modules[__name__], sys.modules.get("primia_syft_compat")):
            if mod is not None:
                mod.hook, mod.local_worker = self, self.local_worker
        torch_module.torch_hooked = True
        if not hasattr(torch_module.Tensor, "tag"):
            def tag(t, *tags):
                cur = set(getattr(t, "_primia_tags", ()))
                cur.update(tags)
                t._primia_tags = cur
                return t

            torch_module.Tensor.tag = tag
            torch_module.Tensor.tags = property(lambda t: getattr(t, "_primia_tags", set()))


class ObjectStore:
    """worker.object_store (syft/generic/object_storage.py): id -> object, searchable by tag."""

    garbage_delay = 0      # (PySyft's deferred deletion of remote objects: nothing to defer here)

    def __init__(self):
        self._objects = {}
        self._next = 0

    def set_obj(self, obj):
        self._objects[self._next] = obj
        self._next += 1
        return self._next - 1

    def clear_objects(self):
        self._objects.clear()

    def find_by_tag(self, tag):
        return [o for o in self._objects.values() if tag in getattr(o, "_primia_tags", ())]

    def __len__(self):
        return len(self._objects)


class VirtualWorker:
    """sy.VirtualWorker(hook, id=..., verbose=False) (syft/workers/virtual.py; base.py:192 load_data, :229 search).
    `device`: the GPU this client's tensors live on (rank k of a one-process-per-GPU launch passes its own device)."""

    def __init__(self, hook=None, id="worker", verbose=False, device=None, data=None):
        self.hook, self.id, self.verbose = hook, id, verbose
        self.device = torch.device(device) if device is not None else None
        self.object_store = ObjectStore()
        if data:
            self.load_data(data)

    def load_data(self, data):
        """base.py:192-207: register every tensor on this worker (moved to the worker's device if it has one)."""
        for t in data:
            tags = getattr(t, "_primia_tags", set())
            if self.device is not None and t.device != self.device:
                t = t.to(self.device)
                t._primia_tags = set(tags)
            self.object_store.set_obj(t)

    def search(self, query):
        """base.py:229-262: objects carrying every tag of the query."""
        tags = [query] if isinstance(query, str) else list(query)
        return [o for o in self.object_store._objects.values()
                if all(t in getattr(o, "_primia_tags", ()) for t in tags)]

    def clear_objects(self):
        self.object_store.clear_objects()
        return self

    def __hash__(self):
        return hash(self.id)

    def __eq__(self, other):
        return self.id == (other.id if isinstance(other, VirtualWorker) else other)

    def __repr__(self):
        return "<VirtualWorker id:{} #objects:{}>".format(self.id, len(self.object_store))


class PrivateGridNetwork:
    """sy.PrivateGridNetwork(*workers) (syft/grid/private_grid.py:24): `search(tag)` -> {worker id: [matches]} over the
    workers that hold a match."""

    def __init__(self, *workers):
        self.workers = list(workers)

    def search(self, *query):
        out = {}
        for w in self.workers:
            hits = w.search(list(query))
            if hits:
                out[w.id] = hits
        return out


class BaseDataset:
    """sy.BaseDataset(data, targets) (syft/frameworks/torch/fl/dataset.py:17)."""

    def __init__(self, data, targets, transform=None):
        self.data, self.targets, self.transform_ = data, targets, transform

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        d = self.data[i]
        return (self.transform_(d) if self.transform_ else d), self.targets[i]


class FederatedDataset:
    """sy.FederatedDataset(datasets) (fl/dataset.py:151): here always the one dataset of one client."""

    def __init__(self, datasets):
        self.datasets = list(datasets)

    @property
    def workers(self):
        return list(range(len(self.datasets)))

    def __len__(self):
        return sum(len(d) for d in self.datasets)


class FederatedDataLoader:
    """sy.FederatedDataLoader(fed_dataset, batch_size, shuffle) (fl/dataloader.py:143) over ONE client's registered
    tensors: yields (data, target) batches that already sit on the client's GPU.  `drop_last` is the reference's
    (default False: the ragged final batch is yielded too — the training loops run it on `engine.sibling(n)` — so every
    sample the class weights and total_L count is trained on)."""

    def __init__(self, federated_dataset, batch_size=8, shuffle=False, num_iterators=1, drop_last=False, seed=0, **kw):
        from .imagefolder import DeviceLoader

        self.federated_dataset = federated_dataset
        ds = federated_dataset.datasets[0]
        self._loader = DeviceLoader(ds.data, ds.targets, batch_size, shuffle, seed, drop_last=drop_last)
        self.drop_last = drop_last
        self.batch_size = batch_size

    @property
    def targets(self):          # held targets: class counts without drawing a batch (datapipe.class_counts)
        return self._loader.targets

    def __len__(self):
        return len(self._loader)

    def __iter__(self):
        return iter(self._loader)


def setup_pysyft(args, hook, verbose=False, device="cuda:0", websockets_config="configs/websetting/config.csv"):
    """torchlib/utils.py:516-860 in the reference's own order of operations, on this module's objects: worker list from
    the CSV (crypto_provider split off), one VirtualWorker per client, every client's mean / std and registered dataset
    tagged and loaded onto its worker, the grid searched for "#traindata" / "#traintargets" / "#datamean" / "#datastd",
    one FederatedDataLoader per worker, the secure average of the statistics.  `args.data_dir`: "synthetic" (seeded
    synthetic shards) or an image-folder tree with worker1..K / validation sub-folders.
    Returns the reference's tuple (train_loader, val_loader, total_L, workers, worker_names, crypto_provider,
    val_mean_std)."""
    from os import path

    from . import fed, imagefolder
    from .torchlib_compat import read_websocket_config

    worker_dict = read_websocket_config(websockets_config)
    worker_names = [w["id"] for w in worker_dict.values()]
    crypto_in_config = "crypto_provider" in worker_names
    assert args.unencrypted_aggregation or crypto_in_config, "No crypto provider in configuration"
    crypto_provider = None
    if crypto_in_config:
        worker_names.remove("crypto_provider")
    if getattr(args, "websockets", False):
        raise NotImplementedError("networked workers are out of scope (SURVEY.md §8f item 4): one node, one GPU per client")
    dev = torch.device(device)
    workers = {n: VirtualWorker(hook, id=n, verbose=False, device=dev) for n in worker_names}
    for w in workers.values():
        w.object_store.clear_objects()
    if not args.unencrypted_aggregation:
        crypto_provider = VirtualWorker(hook, id="crypto_provider", verbose=False, device=dev)
    channels = 3 if args.pretrained else 1
    S, synthetic = args.train_resolution, args.data_dir in (None, "synthetic")
    for i, worker in enumerate(workers.values()):
        if synthetic:
            g = torch.Generator().manual_seed(args.seed + i)
            n = args.batch_size * max(1, 4 - i)
            data = torch.randn(n, channels, S, S, generator=g)
            targets = torch.randint(0, 3, (n,), generator=g)
            mean, std = torch.zeros(channels), torch.ones(channels)
            data, targets = imagefolder.register([data.to(dev)], targets.to(dev), args, 3, args.seed + i)
        else:
            loader, (mean, std) = imagefolder.client_loader(path.join(args.data_dir, "worker{:d}".format(i + 1)), args,
                                                            dev, channels, args.seed + i)
            data, targets = loader.data, loader.targets
        mean.tag("#datamean")
        std.tag("#datastd")
        worker.load_data([mean, std])
        data.tag("#traindata")
        targets.tag("#traintargets")
        worker.load_data([data, targets])
    members = list(workers.values()) + ([crypto_provider] if crypto_provider is not None else [])
    grid = PrivateGridNetwork(*members)
    data, target = grid.search("#traindata"), grid.search("#traintargets")
    train_loader, total_L = {}, 0
    for k, w in enumerate(data.keys()):
        fed_dataset = FederatedDataset([BaseDataset(data[w][0], target[w][0])])
        total_L += len(fed_dataset)
        train_loader[workers[w]] = FederatedDataLoader(fed_dataset, batch_size=args.batch_size, shuffle=True,
                                                       seed=args.seed + k)
    means = [m[0] for m in grid.search("#datamean").values()]
    stds = [s[0] for s in grid.search("#datastd").values()]
    if len(means) != len(workers) or len(stds) != len(workers):
        raise RuntimeError("no datamean/standard deviation was found on (some) worker")
    mean, std = fed.secure_mean_of(list(zip(means, stds)))      # fix_precision().share(...) sums, utils.py:764-794
    val_mean_std = torch.stack([mean.cpu(), std.cpu()])
    if synthetic or not path.isdir(path.join(str(args.data_dir), "validation")):
        g = torch.Generator().manual_seed(args.seed + 999)
        vd = torch.randn(2 * args.batch_size, channels, S, S, generator=g).to(dev)
        vt = torch.randint(0, 3, (2 * args.batch_size,), generator=g).to(dev)
        val_loader = imagefolder.DeviceLoader(vd, vt, args.batch_size, False, 0)
    else:
        val_loader = imagefolder.validation_loader(path.join(args.data_dir, "validation"), args, dev, channels,
                                                   (val_mean_std[0], val_mean_std[1]))
    assert len(train_loader) == len(workers), "data was not correctly loaded"
    if verbose:
        print("Found a total dataset with {:d} samples on remote workers".format(
            sum(len(dl.federated_dataset) for dl in train_loader.values())))
    return train_loader, val_loader, total_L, workers, worker_names, crypto_provider, val_mean_std
