#!/usr/bin/env python3
"""One bf16 training step of the large / huge presets at a small batch (configs 3-5 of BASELINE.json are parity cases, not bench lines):
finite loss, finite gradients, loss close to the fp32 engine on the same inputs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import models_mae
from csmae_hip.optim import FusedAdamW, add_weight_decay

for name, kw, n in (("mae_vit_large_MsLdCeCd", dict(input_size=224, patch_size="16"), 4),
                    ("mae_vit_large_MsLdCeCd", dict(input_size=256, patch_size="16", input_channels=4), 2),
                    ("mae_vit_huge_MsLdCeCd", dict(input_size=224, patch_size="14"), 2)):
    torch.manual_seed(0)
    m = models_mae.__dict__[name](**kw).cuda().train()
    x = torch.randn(n, kw.get("input_channels", 3), kw["input_size"], kw["input_size"], device="cuda")
    losses = {}
    for dt in (torch.float32, torch.bfloat16):
        m.compute_dtype = dt
        m.zero_grad(set_to_none=True)
        torch.manual_seed(1)
        loss, pred, mask = m(x, mask_ratio=0.75)
        loss.backward()
        g = [p.grad for p in m.parameters() if p.grad is not None]
        assert torch.isfinite(loss) and all(torch.isfinite(t).all() for t in g), name
        losses[dt] = float(loss)
    opt = FusedAdamW(add_weight_decay(m, 0.05), lr=1e-4, betas=(0.9, 0.95))
    opt.step()
    torch.cuda.synchronize()
    rel = abs(losses[torch.bfloat16] - losses[torch.float32]) / abs(losses[torch.float32])
    print(f"{name} {kw}: fp32 loss {losses[torch.float32]:.5f}  bf16 loss {losses[torch.bfloat16]:.5f}  rel {rel:.2e}  params {sum(p.numel() for p in m.parameters()) / 1e6:.1f} M")
    assert rel < 2e-2
    del m, opt
    torch.cuda.empty_cache()
print("OK")
