"""The end-to-end trainer loop at the reference config's own store shape (store_batch_size 32 x n_batches_in_buffer 20), folded launches of the
SAE step on and off (tuning key sae_fold): is the leg's time a property of the step or of the box?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit_prisma_amd import _native as N
from vit_prisma_amd.sae.bench_leg import sae_end_to_end_leg

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for fold in (1, 0, 1, 0):
    N.set_tuning("reset")
    N.set_tuning("sae_fold", fold)
    r = sae_end_to_end_leg(dev, steps=40, warmup=24, store_bs=32, n_buf=20)
    print("sae_fold", fold, "ms_per_step", r["ms_per_step"], "tokens/s", r["value"], flush=True)
N.set_tuning("reset")
