"""Fixtures written by the REFERENCE's own classes (build container only):
  sae_ref_ckpt/ref_legacy.pt        StandardSparseAutoencoder.save_model -> {"cfg": <pickled vit_prisma.sae.config.
                                    VisionModelSAERunnerConfig>, "state_dict"} (sae.py:299-320)
  sae_ref_ckpt/weights.pt + config.json   the split form (bare state dict + VisionModelSAERunnerConfig.save_config)
  sae_ref_ckpt/io.pt                an input batch and the reference module's reconstruction of it
  geometric_median.npz              points, the reference's compute_geometric_median(points, maxiter=100).median
                                    (sae/training/geometric_median.py:23-86)
    python tests/golden/gen_golden_sae_ckpt.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from gen_golden_sae import ref_trainer_classes  # noqa: E402

Cfg, SAE, Trainer = ref_trainer_classes()
out = os.path.join(HERE, "sae_ref_ckpt")
os.makedirs(out, exist_ok=True)
cfg = Cfg(d_in=64, expansion_factor=8, activation_fn_str="topk", activation_fn_kwargs={"k": 8}, _device="cpu", _dtype="float32",
          log_to_wandb=False, verbose=False)
assert type(cfg).__module__ == "vit_prisma.sae.config"
torch.manual_seed(3)
m = SAE(cfg)
m.save_model(os.path.join(out, "ref_legacy.pt"))
torch.save(m.state_dict(), os.path.join(out, "weights.pt"))
cfg.save_config(os.path.join(out, "config.json"))
x = torch.randn(5, 64)
torch.save({"x": x, "out": m(x)[0].detach()}, os.path.join(out, "io.pt"))

from vit_prisma.sae.training.geometric_median import compute_geometric_median  # noqa: E402
pts = torch.from_numpy(np.random.RandomState(5).standard_normal((300, 24)).astype(np.float32) * 2.0 + 1.0)
pts[:20] += 15.0                                    # outliers: the median must not follow the mean
res = compute_geometric_median(pts, maxiter=100)
np.savez(os.path.join(HERE, "geometric_median.npz"), points=pts.numpy(), median=res.median.numpy())
print("written", os.listdir(out), float((res.median - pts.mean(0)).norm()))
