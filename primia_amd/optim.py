"""The optimizer object of the host loop: `torch.optim.SGD` / `torch.optim.Adam` as PriMIA constructs them
(train.py:280-303; torchlib/utils.py:1131-1145,1208-1218) bound to one HIP engine.

The arithmetic is `primia_sgd_step` / `primia_adam_step` on the engine's flat arena; this class carries what the
reference's host code touches — `param_groups` (the LearningRateScheduler writes `lr` there, utils.py:84-88),
`zero_grad()`, `step()` — and speaks torch's `state_dict()` / `load_state_dict()` format so that checkpoints
round-trip with the reference (`save_model`, utils.py:1470-1493; resume matrix, train.py:344-389):

    {"state": {i: {"step": n, "exp_avg": T, "exp_avg_sq": T}}, "param_groups": [{"lr", "betas", "eps",
     "weight_decay", "amsgrad", "params": [0..61]}]}

Parameter i is the i-th entry of `named_parameters()` — the arena order.  torch 1.4 keys its saved state by
`id(param)`; like torch's own loader, `load_state_dict` therefore maps saved ids to parameters by POSITION in
`param_groups[*]["params"]`, so files written by torch 1.4 and by torch >= 1.5 both load.
"""
import torch


class EngineOptimizer:
    def __init__(self, engine, kind="SGD", lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        """Constructing an optimizer starts from empty state, exactly like `opt(model.parameters(), **kwargs)`."""
        if kind not in ("SGD", "Adam"):
            raise NotImplementedError("only Adam or SGD supported.")
        self.engine, self.kind = engine, kind
        group = {"lr": lr, "weight_decay": weight_decay}
        if kind == "Adam":
            group.update(betas=tuple(betas), eps=eps, amsgrad=False)
        else:
            group.update(momentum=0, dampening=0, nesterov=False)
        group["params"] = list(range(len(engine.p_entries)))
        self.param_groups = [group]
        engine.reset_optimizer()
        # SGD: step() follows the backward pass directly, so the engine may leave the conv gradients in their
        # accumulators and finish them, the update and the weight refresh in one pass (engine.fuse_sgd_tail)
        engine.fuse_sgd_tail = kind == "SGD"

    @classmethod
    def from_args(cls, engine, args, lr=None):
        """The constructor calls of train.py:280-303 / utils.py:1131-1145."""
        kw = {"lr": args.lr if lr is None else lr, "weight_decay": args.weight_decay}
        if args.optimizer == "Adam":
            kw["betas"] = (args.beta1, args.beta2)
        return cls(engine, args.optimizer, **kw)

    # ---- the three calls of the batch loop ---------------------------------------------------------------------
    def zero_grad(self):
        pass   # every backward pass of the engine overwrites the gradient arena

    def step(self, engine=None):
        """`engine`: the engine whose backward pass produced the gradients, when it is a sibling of the one this optimizer
        was built on (ResNet18Engine.sibling: same parameters and optimizer state, another batch size)."""
        eng = self.engine if engine is None else engine
        if eng is not self.engine and getattr(eng, "_root", eng) is not getattr(self.engine, "_root", self.engine):
            raise ValueError("optimizer.step(engine): not a sibling of the optimizer's engine")
        g = self.param_groups[0]
        if self.kind == "SGD":
            eng.sgd_step(g["lr"], g["weight_decay"])
        else:
            eng.adam_step(g["lr"], g["betas"], g["eps"], g["weight_decay"])

    # ---- torch's checkpoint format --------------------------------------------------------------------------------
    def _slices(self):
        off = 0
        for i, (name, shape) in enumerate(self.engine.p_entries):
            n = int(torch.Size(shape).numel())
            yield i, shape, off, n
            off += n

    def state_dict(self):
        state = {}
        eng = self.engine
        if self.kind == "Adam" and eng.opt_state is not None:
            m, v = eng.opt_state
            for i, shape, off, n in self._slices():
                state[i] = {"step": eng.opt_steps, "exp_avg": m[off:off + n].view(shape).cpu().clone(),
                            "exp_avg_sq": v[off:off + n].view(shape).cpu().clone()}
        return {"state": state, "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        if "param_groups" not in sd or "state" not in sd:
            raise ValueError("not an optimizer state dict")
        saved = sd["param_groups"]
        if len(saved) != 1:
            raise ValueError("loaded state dict has a different number of parameter groups")
        ids = list(saved[0]["params"])
        if len(ids) != len(self.engine.p_entries):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of "
                             "optimizer's group")
        own = self.param_groups[0]["params"]
        self.param_groups[0].update({k: v for k, v in saved[0].items() if k != "params"})
        self.param_groups[0]["params"] = own
        eng = self.engine
        eng.reset_optimizer()
        if not sd["state"]:
            return
        if self.kind != "Adam":
            raise ValueError("the checkpoint carries per-parameter optimizer state, this optimizer (SGD without "
                             "momentum) has none")
        m, v = torch.zeros_like(eng.grads), torch.zeros_like(eng.grads)
        steps = set()
        for (i, shape, off, n), old in zip(self._slices(), ids):
            st = sd["state"].get(old)
            if st is None:       # torch allows parameters without state (never stepped)
                continue
            m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(m))
            v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(v))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("parameters with different step counts cannot share the engine's single counter")
        eng.opt_state = (m, v)
        eng.opt_steps = steps.pop() if steps else 0
