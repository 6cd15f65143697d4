#!/usr/bin/env python3
"""PriMIA-compatible inference CLI (inference.py:47-76, 279-323 of the reference).

    python inference.py --model_weights model_weights/final_*.pt [--data_dir DIR|synthetic]
                        [--encrypted_inference] [--cuda] [--websockets_config CSV] [--http_protocol]

Plain inference runs the HIP engine in eval mode.  `--encrypted_inference` shares the model and
each image between model_owner and data_owner (fixed precision 10^16, protocol "fss", a dealer as
crypto provider) and runs the secret-shared forward of primia_amd.secure, image by image like the
reference's loop.  Output: the reference's JSON on stdout, {"Inference Results": {index: class}}.
"""
import argparse
import json
import os
import sys
from datetime import datetime

import torch

from primia_amd.engine import ResNet18Engine
from primia_amd.secure import Dealer, SecureContext, SecureResNet18
from primia_amd.torchlib_compat import Arguments  # noqa: F401  (checkpoints pickle an Arguments instance)


def load_images(data_dir, n, size, channels, device, mean, std, seed=0, clahe=False):
    """The reference's inference transform (inference.py:176-196: a.Resize(R, R) -> a.CenterCrop(R, R) -> a.ToFloat ->
    a.Normalize with the checkpoint's mean / std) for the first `n` images of `data_dir`, on the GPU through
    primia_image_prepare; RGB or single channel as the checkpoint's stem says (CombinedLoader.change_channels).
    `synthetic` (or no folder given) -> seeded noise; a folder that does not exist is an error."""
    if data_dir in (None, "synthetic"):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(n, channels, size, size, generator=g).to(device)
    if not os.path.isdir(data_dir):
        raise SystemExit("data_dir {!r} does not exist (pass 'synthetic' for seeded noise)".format(data_dir))
    import numpy as np

    from primia_amd import imagefolder
    from primia_amd._lib import call

    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(data_dir) for f in fs
                   if f.lower().endswith(imagefolder.EXTENSIONS) and not f.startswith("._"))[:n]
    if not files:
        raise SystemExit("no images under {!r}".format(data_dir))
    out = torch.empty(len(files), channels, size, size, dtype=torch.float32, device=device)
    m = mean.to(device).float().reshape(-1).contiguous()
    s_ = std.to(device).float().reshape(-1).contiguous()
    if m.numel() != channels:          # a one-element statistic applies to every channel
        m, s_ = m[:1].repeat(channels), s_[:1].repeat(channels)
    if clahe:        # inference.py:182-183: a.CLAHE(always_apply=True, clip_limit=(1, 1)) between the crop and ToFloat
        from primia_amd._lib import query

        wsb = query("primia_clahe_workspace_bytes", size, size, channels)
        ws = torch.empty(wsb, dtype=torch.uint8, device=device)
        u8 = torch.empty(size, size, channels, dtype=torch.uint8, device=device)
    for i, fn in enumerate(files):
        img = torch.from_numpy(np.ascontiguousarray(imagefolder.decode(fn, channels))).to(device)
        if clahe:
            call("primia_image_resize_crop_u8", img, img.shape[0], img.shape[1], channels, size, 0, 0, size, 0, u8)
            call("primia_clahe_u8", u8, size, size, channels, 1.0, ws, wsb, u8)
            call("primia_image_finish", u8, size, channels, m, s_, out[i])
        else:
            call("primia_image_prepare", img, img.shape[0], img.shape[1], channels, size, 0, 0, size, 0, m, s_, out[i])
    return out


if __name__ == "__main__":
    start_time = datetime.now()
    parser = argparse.ArgumentParser()
    parser.add_argument("--data_dir", default=None, help="data to classify")
    parser.add_argument("--model_weights", type=str, required=True, help="model weights to use")
    parser.add_argument("--encrypted_inference", action="store_true", help="Perform encrypted inference")
    parser.add_argument("--websockets_config", default=None, help="accepted for compatibility (in-process parties)")
    parser.add_argument("--cuda", action="store_true", help="Use GPU acceleration (always on here).")
    parser.add_argument("--http_protocol", action="store_true", help="accepted for compatibility")
    parser.add_argument("--num_images", type=int, default=4)
    parser.add_argument("--hip_graph", action="store_true",
                        help="encrypted inference: capture the online phase once as a hipGraph and replay it per image "
                             "(the dealer refills the primitive buffers between images)")
    parser.add_argument("--three_role", action="store_true",
                        help="encrypted inference with model_owner, data_owner and crypto_provider as three ranks "
                             "(launch with `python -m torch.distributed.run --nproc-per-node 3 inference.py ...`): "
                             "one GPU each over RCCL when three are visible, else all on GPU 0 over gloo")
    parser.add_argument("--debug_dealer_seed", type=int, default=None,
                        help="DEBUG ONLY: derive the crypto provider's key from this number (reproducible, hence "
                             "NOT private); by default the key comes from the OS entropy pool and never leaves the "
                             "provider")
    cmd_args = parser.parse_args()
    if cmd_args.debug_dealer_seed is not None:
        print("WARNING: --debug_dealer_seed makes every mask, triple and FSS key predictable: no confidentiality",
              file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("primia_amd runs inference on the GPU only (HIP kernels); no GPU visible")
    device = torch.device("cuda:0")
    state = torch.load(cmd_args.model_weights, map_location="cpu", weights_only=False)
    args = state["args"]
    if not isinstance(args, Arguments):
        args = Arguments.from_namespace(args)
    args.from_previous_checkpoint(cmd_args)
    size = getattr(args, "inference_resolution", args.train_resolution)
    sd = state["model_state_dict"]
    channels = int(sd["conv1.weight"].shape[1])       # 3 for pretrained = yes, else 1 (train.py:262)
    # inference.py:163-174: the checkpoint's statistics, else 0.5 / 0.2
    mean, std = state.get("val_mean_std", (torch.full((channels,), 0.5), torch.full((channels,), 0.2)))
    images = load_images(cmd_args.data_dir, cmd_args.num_images, size, channels, device, mean, std,
                         clahe=bool(getattr(args, "clahe", False)))
    total_pred = []
    if args.encrypted_inference:
        # inference.py:279-286: fix_precision(precision_fractional=16, dtype="long").share(..., protocol="fss")
        if cmd_args.three_role:
            import torch.distributed as dist

            from primia_amd.secure import PartyLink, architecture_of, run_three_role

            multi = torch.cuda.device_count() >= 3
            rank = int(os.environ["RANK"])
            device = torch.device("cuda", rank if multi else 0)
            torch.cuda.set_device(device)
            dist.init_process_group(os.environ.get("PRIMIA_PARTY_BACKEND", "nccl" if multi else "gloo"))
            link = PartyLink(device)
            # one checkpoint file serves all three ranks of a single-node launch; each role is handed only what
            # it owns: the weights (party 0), the images (party 1), the architecture (everyone)
            logits = run_three_role(link, architecture_of(sd), size, images.shape[0],
                                    state_dict=sd if link.role == 0 else None,
                                    images=images.to(device) if link.role == 1 else None,   # (this rank's GPU)
                                    seed=cmd_args.debug_dealer_seed)
            dist.barrier()
            dist.destroy_process_group()
            if link.role != 1:
                sys.exit(0)
            total_pred = [int(o.argmax(dim=1).item()) for o in logits]
            if os.environ.get("PRIMIA_DUMP_LOGITS"):
                torch.save(torch.cat(logits).cpu(), os.environ["PRIMIA_DUMP_LOGITS"])
        elif cmd_args.hip_graph:
            from primia_amd.secure import GraphedSecureInference

            model = GraphedSecureInference(sd, device, input_size=size, precision_fractional=16,
                                           seed=cmd_args.debug_dealer_seed)
        else:
            ctx = SecureContext(Dealer(device, seed=cmd_args.debug_dealer_seed), base=10, precision_fractional=16)
            model = SecureResNet18(ctx, sd, input_size=size)
        logits = []
        for i in range(0 if cmd_args.three_role else images.shape[0]):
            out = model(images[i:i + 1]).clone()      # (the graphed form returns its static output buffer: keep a copy)
            logits.append(out)
            total_pred.append(int(out.argmax(dim=1).item()))
        if logits and os.environ.get("PRIMIA_DUMP_LOGITS"):
            torch.save(torch.cat(logits).cpu(), os.environ["PRIMIA_DUMP_LOGITS"])
    else:
        eng = ResNet18Engine(1, sd["fc.weight"].shape[0], sd["conv1.weight"].shape[1], size,
                             getattr(args, "pooling_type", "max"), dtype=torch.float32, device=device)
        eng.load_state_dict(sd)
        eng.eval()
        for i in range(images.shape[0]):
            total_pred.append(int(eng.forward(images[i:i + 1]).argmax(dim=1).item()))
    print(json.dumps({"Inference Results": {i: p for i, p in enumerate(total_pred)}}))
    print("Took {:s} seconds.".format(str(datetime.now() - start_time)), file=sys.stderr)
