"""Golden on-disk activation cache (build container only): the REFERENCE's
VisionActivationsStore.generate_cached_activations_from_dataset writes the {idx}.pt shards of the tiny model's
blocks.1.hook_resid_post, and the reference's CacheVisionActivationStore reads them back; both results are
committed (tests/golden/act_cache_tiny/) so that the format can be checked on machines without the reference."""
import os, shutil, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from gen_golden_vit import build_reference_model
from vit_prisma_amd.synth import synth_images
import importlib.util


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join("/root/reference/src/vit_prisma", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m


model, arch = build_reference_model("tiny")
store_mod = _load("vit_prisma.sae.training.activations_store", "sae/training/activations_store.py")
out_dir = os.path.join(HERE, "act_cache_tiny")
shutil.rmtree(out_dir, ignore_errors=True)


class Cfg:                      # the fields the two reference classes read
    cached_activations_path = out_dir
    store_batch_size = 2
    num_workers = 0
    device = "cpu"
    dtype = torch.float32
    context_size = 17
    hook_point_layer = 1
    hook_point = "blocks.1.hook_resid_post"
    hook_point_head_index = None
    cls_token_only = False
    use_patches_only = False
    d_in = 64
    n_batches_in_buffer = 4
    train_batch_size = 16
    use_cached_activations = True
    is_transcoder = False


imgs = torch.from_numpy(synth_images(arch, 7, 9))
ds = torch.utils.data.TensorDataset(imgs, torch.zeros(7, dtype=torch.long))
st = store_mod.VisionActivationsStore.__new__(store_mod.VisionActivationsStore)
st.cfg, st.model, st.dataset = Cfg, model, ds
with torch.no_grad():
    st.generate_cached_activations_from_dataset(tokens_per_file=50)
files = sorted(os.listdir(out_dir))
print(files, [tuple(torch.load(os.path.join(out_dir, f)).shape) for f in files])
torch.manual_seed(0)
rd = store_mod.CacheVisionActivationStore.__new__(store_mod.CacheVisionActivationStore)
rd.cfg = Cfg
buf = rd.get_buffer(2)          # 2 store batches = 4 images = 68 tokens -> shard 0 + part of shard 1
np.savez_compressed(os.path.join(out_dir, "reference_reader.npz"), buffer_2_batches=buf.numpy(),
                    images=imgs.numpy())
print("buffer", tuple(buf.shape))
