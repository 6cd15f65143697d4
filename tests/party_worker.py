"""One rank of the three-role encrypted-inference test (launched by test_gpu_secure.py through
torch.distributed.run, 3 ranks sharing GPU 0 over gloo).  Writes the data owner's decoded logits."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primia_amd.secure import PartyLink, architecture_of, run_three_role  # noqa: E402
from tests.test_gpu_secure import mini_state_dict  # noqa: E402

if __name__ == "__main__":
    out_path, pf = sys.argv[1], int(sys.argv[2])
    device = torch.device("cuda:0")
    dist.init_process_group("gloo")
    gen = torch.Generator().manual_seed(21)
    sd = mini_state_dict(gen)
    images = torch.randn(2, 3, 16, 16, generator=gen)
    blocks = [("layer1.0", 1), ("layer2.0", 2)]
    link = PartyLink(device)
    res = run_three_role(link, architecture_of(sd), 16, 2, state_dict=sd if link.role == 0 else None,
                         images=images.to(device) if link.role == 1 else None, seed=5, blocks=blocks,
                         precision_fractional=pf)
    if link.role in (0, 1):
        torch.save(torch.cat(res).cpu(), f"{out_path}.{link.role}")
    dist.barrier()
    dist.destroy_process_group()
