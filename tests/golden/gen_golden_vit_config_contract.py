"""The reference's HookedViTConfig (configs/HookedViTConfig.py) as a contract (build container only): every dataclass field of a default
instance and of the two target architectures' keyword sets after __post_init__.

    python tests/golden/gen_golden_vit_config_contract.py     ->  tests/golden/vit_config_contract.json"""
import dataclasses
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from vit_prisma_amd.synth import ARCHS  # noqa: E402


def plain(v):
    if isinstance(v, (torch.dtype, torch.device)):
        return str(v)
    if isinstance(v, (list, tuple)):
        return [plain(t) for t in v]
    if isinstance(v, dict):
        return {str(k): plain(t) for k, t in v.items()}
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    return repr(type(v))


def contract(Cfg):
    res = {}
    for tag, kw in (("defaults", {}), ("clip-vit-b32", ARCHS["clip-vit-b32"]), ("clip-vit-l14-336", ARCHS["clip-vit-l14-336"]), ("tiny", ARCHS["tiny"])):
        cfg = Cfg(**kw, device="cpu")
        res[tag] = {f.name: plain(getattr(cfg, f.name, "<unset>")) for f in dataclasses.fields(cfg)}
    return res


if __name__ == "__main__":
    from _refimport import reference_classes
    reference_classes()                                            # (installs the import stubs the reference needs here)
    from vit_prisma.configs.HookedViTConfig import HookedViTConfig
    res = contract(HookedViTConfig)
    with open(os.path.join(HERE, "vit_config_contract.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in res.items()})
