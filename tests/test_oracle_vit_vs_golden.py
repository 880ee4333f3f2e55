"""Pin the numpy oracle (oracle/vit_oracle.py) against golden fixtures produced by executing the
reference itself (tests/golden/gen_golden_vit.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle.vit_oracle import fingerprint, hook_names_in_order, vit_forward
from vit_prisma_amd.synth import ARCHS, synth_images, synth_vit_state

from conftest import GOLDEN, rel_fro

FP32_TOL = 2e-5      # rel-Frobenius, fp32 oracle vs fp32 reference (different summation orders)


def _check_fp(name, got, want_fp, tol=FP32_TOL):
    got_fp = fingerprint(got)
    assert got_fp["shape"] == want_fp["shape"], name
    scale = max(want_fp["l2"], 1e-30)
    vals_g, vals_w = np.array(got_fp["vals"]), np.array(want_fp["vals"])
    n = float(np.prod(want_fp["shape"]))
    rms = scale / np.sqrt(n)
    assert abs(got_fp["l2"] - want_fp["l2"]) <= tol * scale, name
    # sums are cancellation-prone: bound by tol * l2 * sqrt(n) worst case, use l2*sqrt(n)*tol/8
    assert abs(got_fp["sum"] - want_fp["sum"]) <= tol * scale * np.sqrt(n), name
    assert abs(got_fp["wsum"] - want_fp["wsum"]) <= tol * scale * np.sqrt(n), name
    assert np.max(np.abs(vals_g - vals_w)) <= 50 * tol * max(rms, np.max(np.abs(vals_w))), name


@pytest.mark.parametrize("arch_name,bs,fname", [
    ("tiny", 3, "vit_tiny_full.npz"),
    ("tiny-ragged", 2, "vit_tiny_ragged_full.npz"),
])
def test_oracle_matches_reference_full_tensors(arch_name, bs, fname):
    g = np.load(os.path.join(GOLDEN, fname))
    arch = ARCHS[arch_name]
    out, cache = vit_forward(synth_vit_state(arch, 0), arch, synth_images(arch, bs, 1))
    keys = [str(k) for k in g["__keys__"]]
    assert list(cache.keys()) == keys                      # bit-exact hook order
    assert keys == hook_names_in_order(arch)
    for k in keys:
        assert cache[k].shape == g[k].shape, k
        assert rel_fro(cache[k], g[k]) < FP32_TOL, k
    assert rel_fro(out, g["__out__"]) < FP32_TOL


def test_oracle_matches_reference_b32_bs16():
    with open(os.path.join(GOLDEN, "vit_b32_fp32_bs16.json")) as f:
        G = json.load(f)
    arch = ARCHS["clip-vit-b32"]
    sd = synth_vit_state(arch, 0)
    imgs = synth_images(arch, 16, 1)
    out, cache = vit_forward(sd, arch, imgs)
    want = G["all"]
    assert list(cache.keys()) == want["keys"]
    assert len(want["keys"]) == 214                         # 6 + 17*12 + 4
    for k in want["keys"]:
        _check_fp(k, cache[k], want["cache"][k])
    _check_fp("out", out, want["out"])

    # harvest form: names_filter list + stop_at_layer
    w = G["stop7_filter"]
    out7, c7 = vit_forward(sd, arch, imgs, stop_at_layer=7, names_filter=["blocks.6.hook_resid_post"])
    assert list(c7.keys()) == w["keys"] == ["blocks.6.hook_resid_post"]
    _check_fp("stop7", c7["blocks.6.hook_resid_post"], w["cache"]["blocks.6.hook_resid_post"])
    _check_fp("stop7_out", out7, w["out"])

    # negative stop_at_layer
    w = G["stop_neg9_bs2"]
    outm, cm = vit_forward(sd, arch, imgs[:2], stop_at_layer=-9)
    assert list(cm.keys()) == w["keys"]
    _check_fp("stopneg_out", outm, w["out"])

    # callable filter
    w = G["callable_bs2"]
    _, cc = vit_forward(sd, arch, imgs[:2], names_filter=lambda n: n.endswith("hook_pattern") or n == "hook_embed")
    assert list(cc.keys()) == w["keys"]
    for k in w["keys"]:
        _check_fp(k, cc[k], w["cache"][k])


@pytest.mark.slow
def test_oracle_matches_reference_l14_patterns():
    with open(os.path.join(GOLDEN, "vit_l14_fp32_bs1.json")) as f:
        G = json.load(f)
    arch = ARCHS["clip-vit-l14-336"]
    want = G["sel"]
    out, cache = vit_forward(synth_vit_state(arch, 0), arch, synth_images(arch, 1, 1),
                             names_filter=want["keys"])
    assert list(cache.keys()) == want["keys"]
    for k in want["keys"]:
        _check_fp(k, cache[k], want["cache"][k])
    _check_fp("out", out, want["out"])
    assert G["all_keys"] == hook_names_in_order(arch)
    assert len(G["all_keys"]) == 418                        # 6 + 17*24 + 4


def test_oracle_matches_reference_on_the_massive_activation_state():
    """Round 6: the oracle against the reference's own fp32 run on synth_vit_state(outliers=True) -- a residual
    stream with channels at |x| ~ 100 beside an rms of ~1.2 (tests/golden/vit_b32_outliers_bf16_budget.json) -- all 214 keys."""
    with open(os.path.join(GOLDEN, "vit_b32_outliers_bf16_budget.json")) as f:
        G = json.load(f)
    assert G["outliers"] and G["resid6_absmax_over_rms"] > 15.0
    arch = ARCHS["clip-vit-b32"]
    out, cache = vit_forward(synth_vit_state(arch, 0, outliers=True), arch, synth_images(arch, G["batch"], G["seed"]))
    want = G["fp32"]
    assert list(cache.keys()) == want["keys"] and len(want["keys"]) == 214
    for k in want["keys"]:
        _check_fp(k, cache[k], want["cache"][k])
    _check_fp("out", out, want["out"])
