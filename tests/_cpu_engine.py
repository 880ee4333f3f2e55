"""A CPU stand-in with NativeSAE's interface, built from the oracle (test infrastructure only).

The data-parallel orchestration of ``VisionSAETrainer._native_dp_step`` -- which collective moves what, which rows a rank
clips / projects / updates, when the parameter all-gathers are waited for -- is plain host code around the engine's
methods.  This twin lets that code run under gloo on CPU (world 2 and 4) where the HIP engine cannot."""
from typing import Optional

import numpy as np
import torch

from oracle import sae_oracle as O


class OracleEngine:
    def __init__(self, sae, k: int, max_tokens: int):
        self.params = {n: getattr(sae, n).data for n in ("W_enc", "W_dec", "b_enc", "b_dec")}
        self.d_in, self.d_sae = self.params["W_enc"].shape
        self.k, self.max_tokens = k, max_tokens
        nW = self.d_in * self.d_sae
        self.n_flat = 2 * nW + self.d_sae + self.d_in
        z = lambda: torch.zeros(self.n_flat)                  # noqa: E731

        def views(flat):
            return dict(W_encT=flat[:nW].view(self.d_sae, self.d_in), W_dec=flat[nW:2 * nW].view(self.d_sae, self.d_in),
                        b_enc=flat[2 * nW:2 * nW + self.d_sae], b_dec=flat[2 * nW + self.d_sae:])

        self.flat_g, self.flat_m, self.flat_v = z(), z(), z()
        self._g, self._m, self._v = views(self.flat_g), views(self.flat_m), views(self.flat_v)
        self.W_encT = self.params["W_enc"].t().contiguous()
        self.fire_count = torch.zeros(self.d_sae)
        self.scalars = torch.zeros(8)
        self.act_freq_scores = torch.zeros(self.d_sae)
        self.n_fwd_since_fired = torch.zeros(self.d_sae)
        self.adam_step = 0
        self.filtered_encoder = False
        self.lazy_w_enc = False
        self.sae_out = torch.zeros(max_tokens, self.d_in)
        self._last = None                                         # (batch_mean, n_global) of the last top-k step, for topk_ghost

    def materialize_w_enc(self):                                  # (NativeSAE.lazy_w_enc: this twin keeps W_enc current)
        pass

    def invalidate(self):
        pass

    def _P(self):
        return {n: t.numpy() for n, t in self.params.items()}          # (numpy views of the parameter storage)

    def renorm_decoder(self):
        O.renorm_decoder(self._P())

    def step(self, x, batch_mean=None, n_global=None, update_stats=True, want_out=False, renorm_decoder=False, sparse_grads=False):
        if renorm_decoder:
            self.renorm_decoder()
        P, xn = self._P(), x.numpy()
        bm = None if batch_mean is None else batch_mean.numpy().astype(np.float32)
        fw = O.sae_forward(P, xn, self.k, batch_mean=bm, n_global=n_global)
        g = O.sae_backward(P, xn, fw, n_global=n_global)
        self._g["W_encT"].copy_(torch.from_numpy(g["W_enc"].T.copy()))
        for n in ("W_dec", "b_enc", "b_dec"):
            self._g[n].copy_(torch.from_numpy(g[n]))
        fire = (fw["feature_acts"] > 0).sum(axis=0).astype(np.float32)
        self.fire_count.copy_(torch.from_numpy(fire))
        self.scalars[0], self.scalars[1], self.scalars[2] = float(fw["loss"]), float(fw["mse_loss"]), float(fw["l0"])
        self.sae_out[:x.shape[0]].copy_(torch.from_numpy(fw["sae_out"]))
        self._last = (bm, n_global)
        if update_stats:
            self.act_freq_scores += self.fire_count
            self.n_fwd_since_fired += 1
            self.n_fwd_since_fired[self.fire_count > 0] = 0

    @staticmethod
    def _ghost_global(gg):
        """NativeSAE's ghost_global = (mean of sae_out - x over the global batch, its mse loss, n_global) -> the oracle's form"""
        return None if gg is None else (-gg[0].numpy().astype(np.float32), float(gg[1].reshape(-1)[0]))

    def topk_ghost(self, x, dead_mask, ghost_global=None):
        """The twin of NativeSAE.topk_ghost (pv_sae_topk_ghost): the step's gradients with the ghost term's added."""
        P, xn = self._P(), x.numpy()
        bm, ng = self._last
        fw = O.sae_forward(P, xn, self.k, batch_mean=bm, n_global=ng, dead_mask=dead_mask.numpy().astype(bool),
                           ghost_global=self._ghost_global(ghost_global))
        g = O.sae_backward(P, xn, fw, n_global=ng)
        self._g["W_encT"].copy_(torch.from_numpy(g["W_enc"].T.copy()))
        for n in ("W_dec", "b_enc", "b_dec"):
            self._g[n].copy_(torch.from_numpy(g[n]))
        self.scalars[0], self.scalars[5] = float(fw["loss"]), float(fw["ghost_loss"])

    def dense_step(self, x, l1_coefficient, batch_mean=None, n_global=None, update_stats=True, want_out=False, renorm_decoder=True,
                   dead_mask=None, target=None, ghost_global=None):
        """The twin of NativeSAE.dense_step (pv_sae_dense_step): the ReLU + L1 form of the oracle (+ ghost gradients)."""
        assert target is None
        if renorm_decoder:
            self.renorm_decoder()
        P, xn = self._P(), x.numpy()
        bm = None if batch_mean is None else batch_mean.numpy().astype(np.float32)
        fw = O.sae_forward(P, xn, None, batch_mean=bm, n_global=n_global, l1_coefficient=l1_coefficient,
                           dead_mask=None if dead_mask is None else dead_mask.numpy().astype(bool),
                           ghost_global=self._ghost_global(ghost_global))
        g = O.sae_backward(P, xn, fw, n_global=n_global, l1_coefficient=l1_coefficient)
        self._g["W_encT"].copy_(torch.from_numpy(g["W_enc"].T.copy()))
        for n in ("W_dec", "b_enc", "b_dec"):
            self._g[n].copy_(torch.from_numpy(g[n]))
        self.fire_count.copy_(torch.from_numpy((fw["feature_acts"] > 0).sum(axis=0).astype(np.float32)))
        self.scalars[0], self.scalars[1], self.scalars[2] = float(fw["loss"]), float(fw["mse_loss"]), float(fw["l0"])
        self.scalars[4] = float(fw["l1_loss"])
        self.scalars[5] = 0.0 if fw["ghost_loss"] is None else float(fw["ghost_loss"])
        self.sae_out[:x.shape[0]].copy_(torch.from_numpy(fw["sae_out"]))
        if update_stats:
            self.act_freq_scores += self.fire_count
            self.n_fwd_since_fired += 1
            self.n_fwd_since_fired[self.fire_count > 0] = 0

    transcoder = False

    def relu_step(self, x, l1_coefficient, cap=None, **kw):
        """The twin of NativeSAE.relu_step (pv_sae_relu_step): sparse or dense is the kernels' business -- the same step either way."""
        kw.pop("sparse_grads", None)
        return self.dense_step(x, l1_coefficient, **kw)

    def grad_sqnorm(self, from_step=False):
        self.scalars[3] = float((self.flat_g.double() ** 2).sum())

    def grad_sqnorm_rows(self, j_lo, j_hi, include_b_dec):
        s = sum(float((self._g[n][j_lo:j_hi].double() ** 2).sum()) for n in ("W_encT", "W_dec", "b_enc"))
        if include_b_dec:
            s += float((self._g["b_dec"].double() ** 2).sum())
        self.scalars[3] = s

    def apply(self, lr, max_grad_norm, j_lo=0, j_hi=None):
        j_hi = self.d_sae if j_hi is None else j_hi
        self.adam_step += 1
        total = float(self.scalars[3]) ** 0.5
        coef = min(max_grad_norm / (total + 1e-6), 1.0) if max_grad_norm else 1.0
        sl = slice(j_lo, j_hi)
        rows = {"W_encT": (self.W_encT, sl), "W_dec": (self.params["W_dec"], sl), "b_enc": (self.params["b_enc"], sl),
                "b_dec": (self.params["b_dec"], slice(None))}
        P, g, m, v = {}, {}, {}, {}
        for n, (w, s_) in rows.items():
            P[n] = w[s_].numpy()
            g[n] = (self._g[n][s_] * coef).numpy().copy()
            m[n], v[n] = self._m[n][s_].numpy(), self._v[n][s_].numpy()
        par = (g["W_dec"] * P["W_dec"]).sum(axis=1, keepdims=True)      # remove_gradient_parallel_to_decoder_directions
        g["W_dec"] -= par * P["W_dec"]
        O.adam_step(P, g, m, v, lr, self.adam_step)
        self.params["W_enc"][:, sl] = self.W_encT[sl].t()               # the engine keeps W_enc in step with W_encT

    def sync_shadows(self, from_transposed=False, j_lo=0, j_hi=None):
        j_hi = self.d_sae if j_hi is None else j_hi
        if from_transposed:
            self.params["W_enc"][:, j_lo:j_hi] = self.W_encT[j_lo:j_hi].t()
        else:
            self.W_encT[j_lo:j_hi] = self.params["W_enc"][:, j_lo:j_hi].t()

    def fallback_rows(self):
        return 0


class OracleShardEngine(OracleEngine):
    """The twin of a NativeSAE built over ONE RANK'S FEATURE SHARD, with the cut-at-the-reconstruction entry points of the
    feature-parallel step (pv_sae_encode_topk / pv_sae_tp_partial / pv_sae_tp_finish; vit_prisma_amd/sae/feature_parallel.py)."""

    def __init__(self, W_enc, W_dec, b_enc, b_dec, k: int, max_tokens: int):
        import types
        holder = types.SimpleNamespace(**{n: types.SimpleNamespace(data=t) for n, t in
                                          dict(W_enc=W_enc, W_dec=W_dec, b_enc=b_enc, b_dec=b_dec).items()})
        super().__init__(holder, k, max_tokens)
        self.g = dict(W_enc=self._g["W_encT"], W_dec=self._g["W_dec"], b_enc=self._g["b_enc"], b_dec=self._g["b_dec"])
        self._enc = None

    # ---- the exchange buffers of the feature-parallel step (NativeSAE.tp_bind / tp_merge / tp_bucket_*) ----
    def tp_bind(self, pack, bucket, lo, d_sae_total):
        self._pack, self._tp = pack, (int(lo), int(d_sae_total))
        self._g["b_dec"] = self.g["b_dec"] = bucket[:self.d_in]
        self.fire_count = bucket[self.d_in + 4 + lo:self.d_in + 4 + lo + self.d_sae]

    def tp_merge(self, gathered, world, rank, n):
        """The host-side statement of pv_sae_tp_merge: rank all W k candidates of a token by (value desc, GLOBAL feature index
        asc) with two stable sorts, keep the first k."""
        k = self.k
        v = gathered[:, 0].view(torch.float32).permute(1, 0, 2).reshape(n, world * k)
        g = (gathered[:, 1] + torch.arange(world, dtype=torch.int32).view(world, 1, 1) * self.d_sae).permute(1, 0, 2).reshape(n, world * k).long()
        o1 = torch.argsort(g, dim=1, stable=True)
        o2 = torch.argsort(torch.gather(v, 1, o1), dim=1, descending=True, stable=True)
        order = torch.gather(o1, 1, o2)
        keep = torch.zeros(n, world * k, dtype=torch.bool)
        keep.scatter_(1, order[:, :k], True)
        mine = keep[:, rank * k:(rank + 1) * k]
        return torch.where(mine, gathered[rank, 0].view(torch.float32), torch.zeros(()))

    def tp_bucket_pack(self, bucket):
        lo, total = self._tp
        d = self.d_in
        bucket[d] = sum(float((self._g[n].double() ** 2).sum()) for n in ("W_encT", "W_dec", "b_enc"))
        bucket[d + 1] = self.scalars[2]
        bucket[d + 2:d + 4] = 0
        fire = bucket[d + 4:]
        fire[:lo] = 0
        fire[lo + self.d_sae:] = 0

    def tp_bucket_unpack(self, bucket):
        d = self.d_in
        self.scalars[3] = float(bucket[d]) + float((bucket[:d].double() ** 2).sum())
        self.scalars[2] = bucket[d + 1]

    def encode_topk(self, x, want_ln_stats=True):
        out = self._encode_topk(x)
        if getattr(self, "_pack", None) is not None:
            n = x.shape[0]
            self._pack[0, :n].view(torch.float32).copy_(out[1])
            self._pack[1, :n].copy_(out[0])
        return out

    def _encode_topk(self, x):
        P, xn = self._P(), x.numpy()
        xh, mu, std = O.ln_in(xn)
        sae_in = xh - P["b_dec"]
        pre = sae_in @ P["W_enc"] + P["b_enc"]
        idx, vals = O.topk_mask(pre, self.k)
        nf = np.sqrt(((xn - xn.mean(axis=0, keepdims=True)) ** 2).sum(axis=-1, keepdims=True))
        self._enc = dict(sae_in=sae_in, mu=mu, std=std, nf=nf)
        return (torch.from_numpy(idx.astype(np.int32)), torch.from_numpy(vals.astype(np.float32)),
                torch.from_numpy(mu[:, 0].copy()), torch.from_numpy(std[:, 0].copy()))

    def tp_partial(self, idx, val, renorm_decoder=True):
        if renorm_decoder:
            self.renorm_decoder()
        W_dec = self._P()["W_dec"]
        i, v = idx.numpy().astype(np.int64), val.numpy()
        return torch.from_numpy(np.einsum("nk,nkd->nd", v, W_dec[i]).astype(np.float32))

    def tp_finish(self, x, pre_sum, idx, val, n_global=None, enc_term_only=False, update_stats=False):
        P, xn, e = self._P(), x.numpy(), self._enc
        N, d = xn.shape
        ng = N if n_global is None else n_global
        dt = np.float32
        sae_out = (pre_sum.numpy() + P["b_dec"]) * e["std"] + e["mu"]
        mse = ((sae_out - xn) ** 2 / e["nf"]).sum() / dt(ng * d)
        d_pre = dt(2.0) * (sae_out - xn) / e["nf"] / dt(ng * d) * e["std"]
        feats = np.zeros((N, self.d_sae), dt)
        np.put_along_axis(feats, idx.numpy().astype(np.int64), np.maximum(val.numpy(), 0), axis=-1)
        d_hidden = np.where(feats > 0, d_pre @ P["W_dec"].T, dt(0))
        self._g["W_dec"].copy_(torch.from_numpy(feats.T @ d_pre))
        self._g["W_encT"].copy_(torch.from_numpy((e["sae_in"].T @ d_hidden).T.copy()))
        self._g["b_enc"].copy_(torch.from_numpy(d_hidden.sum(axis=0)))
        gb = -(d_hidden @ P["W_enc"].T).sum(axis=0)
        if not enc_term_only:
            gb = gb + d_pre.sum(axis=0)
        self._g["b_dec"].copy_(torch.from_numpy(gb.astype(dt)))
        self.fire_count.copy_(torch.from_numpy((feats > 0).sum(axis=0).astype(dt)))
        self.scalars[0] = self.scalars[1] = float(mse)
        self.scalars[2] = float((feats > 0).sum()) / N


class OracleGatedEngine(OracleEngine):
    """The twin of a NativeSAE built with ``gated=...`` (pv_sae_gated_step): the oracle's gated form."""
    gated = True

    def __init__(self, sae, max_tokens: int):
        super().__init__(sae, 1, max_tokens)
        for n in ("b_gate", "r_mag", "b_mag"):
            self.params[n] = getattr(sae, n).data
        nW = self.d_in * self.d_sae
        base = 2 * nW + self.d_sae + self.d_in
        self.n_flat = base + 3 * self.d_sae
        self.flat_g, self.flat_m, self.flat_v = (torch.zeros(self.n_flat) for _ in range(3))

        def views(flat):
            v = dict(W_encT=flat[:nW].view(self.d_sae, self.d_in), W_dec=flat[nW:2 * nW].view(self.d_sae, self.d_in),
                     b_enc=flat[2 * nW:2 * nW + self.d_sae], b_dec=flat[2 * nW + self.d_sae:base])
            for i, n in enumerate(("b_gate", "r_mag", "b_mag")):
                v[n] = flat[base + i * self.d_sae:base + (i + 1) * self.d_sae]
            return v

        self._g, self._m, self._v = views(self.flat_g), views(self.flat_m), views(self.flat_v)

    gated_topk = False                                            # (set by the test: the top-k form, NativeSAE(gated_topk=True))

    def gated_topk_step(self, x, batch_mean=None, n_global=None, update_stats=True, want_out=False):
        assert self.gated_topk
        return self.gated_step(x, 0.0, batch_mean=batch_mean, n_global=n_global, update_stats=update_stats, want_out=want_out, k=self.k)

    def gated_step(self, x, l1_coefficient, batch_mean=None, n_global=None, update_stats=True, want_out=False, cap=None, sparse=True,
                   k=None):
        self.renorm_decoder()
        P = {n: t.numpy() for n, t in self.params.items() if n != "b_enc"}
        xn = x.numpy()
        bm = None if batch_mean is None else batch_mean.numpy().astype(np.float32)
        fw = O.gated_forward(P, xn, l1_coefficient=l1_coefficient, batch_mean=bm, n_global=n_global, k=k)
        g = O.gated_backward(P, xn, fw, l1_coefficient=l1_coefficient)
        self.flat_g.zero_()
        self._g["W_encT"].copy_(torch.from_numpy(g["W_enc"].T.copy()))
        for n in ("W_dec", "b_dec", "b_gate", "r_mag", "b_mag"):
            self._g[n].copy_(torch.from_numpy(g[n]))
        self.fire_count.copy_(torch.from_numpy((fw["feature_acts"] > 0).sum(axis=0).astype(np.float32)))
        sc = self.scalars
        sc[0], sc[1], sc[2], sc[4], sc[6] = float(fw["loss"]), float(fw["mse_loss"]), float(fw["l0"]), float(fw["l1_loss"]), float(fw["aux_loss"])
        if update_stats:
            self.act_freq_scores += self.fire_count
            self.n_fwd_since_fired += 1
            self.n_fwd_since_fired[self.fire_count > 0] = 0

    def apply(self, lr, max_grad_norm, j_lo=0, j_hi=None):
        assert j_lo == 0 and j_hi in (None, self.d_sae)
        self.adam_step += 1
        total = float(self.scalars[3]) ** 0.5
        coef = min(max_grad_norm / (total + 1e-6), 1.0) if max_grad_norm else 1.0
        names = ("W_encT", "W_dec", "b_dec", "b_gate", "r_mag", "b_mag")                # (b_enc: no gradient, untouched)
        W = {"W_encT": self.W_encT, **{n: self.params[n] for n in names[1:]}}
        P = {n: W[n].numpy() for n in names}
        g = {n: (self._g[n] * coef).numpy().copy() for n in names}
        m = {n: self._m[n].numpy() for n in names}
        v = {n: self._v[n].numpy() for n in names}
        par = (g["W_dec"] * P["W_dec"]).sum(axis=1, keepdims=True)
        g["W_dec"] -= par * P["W_dec"]
        O.adam_step(P, g, m, v, lr, self.adam_step)
        self.params["W_enc"].copy_(self.W_encT.t())



class OracleTranscoderEngine(OracleEngine):
    """The twin of a NativeSAE built with ``b_dec_out`` / ``W_skip`` (pv_sae_state.tc) on its top-k step: the oracle's transcoder form
    (target activation, the decoder's own bias, the skip matrix)."""
    transcoder = True

    def __init__(self, sae, k: int, max_tokens: int):
        super().__init__(sae, k, max_tokens)
        self.extra = [n for n in ("b_dec_out", "W_skip") if getattr(sae, n, None) is not None]
        for n in self.extra:
            self.params[n] = getattr(sae, n).data
        nW = self.d_in * self.d_sae
        base = 2 * nW + self.d_sae + self.d_in
        sizes = {"b_dec_out": self.d_in, "W_skip": self.d_in * self.d_in}
        self.n_flat = base + sum(sizes[n] for n in self.extra)
        self.flat_g, self.flat_m, self.flat_v = (torch.zeros(self.n_flat) for _ in range(3))

        def views(flat):
            v = dict(W_encT=flat[:nW].view(self.d_sae, self.d_in), W_dec=flat[nW:2 * nW].view(self.d_sae, self.d_in),
                     b_enc=flat[2 * nW:2 * nW + self.d_sae], b_dec=flat[2 * nW + self.d_sae:base])
            off = base
            for n in self.extra:
                v[n] = flat[off:off + sizes[n]].view(self.params[n].shape)
                off += sizes[n]
            return v

        self._g, self._m, self._v = views(self.flat_g), views(self.flat_m), views(self.flat_v)

    def step(self, x, batch_mean=None, n_global=None, update_stats=True, want_out=False, renorm_decoder=False, sparse_grads=False,
             target=None):
        assert target is not None
        if renorm_decoder:
            self.renorm_decoder()
        P, xn, yn = self._P(), x.numpy(), target.numpy()
        bm = None if batch_mean is None else batch_mean.numpy().astype(np.float32)
        fw = O.sae_forward(P, xn, self.k, batch_mean=bm, n_global=n_global, target=yn)
        g = O.sae_backward(P, xn, fw, n_global=n_global)
        self._g["W_encT"].copy_(torch.from_numpy(g["W_enc"].T.copy()))
        for n in ["W_dec", "b_enc", "b_dec"] + self.extra:
            self._g[n].copy_(torch.from_numpy(g[n]))
        self.fire_count.copy_(torch.from_numpy((fw["feature_acts"] > 0).sum(axis=0).astype(np.float32)))
        self.scalars[0], self.scalars[1], self.scalars[2] = float(fw["loss"]), float(fw["mse_loss"]), float(fw["l0"])
        if update_stats:
            self.act_freq_scores += self.fire_count
            self.n_fwd_since_fired += 1
            self.n_fwd_since_fired[self.fire_count > 0] = 0

    def apply(self, lr, max_grad_norm, j_lo=0, j_hi=None):
        assert j_lo == 0 and j_hi in (None, self.d_sae)
        self.adam_step += 1
        total = float(self.scalars[3]) ** 0.5
        coef = min(max_grad_norm / (total + 1e-6), 1.0) if max_grad_norm else 1.0
        names = ["W_encT", "W_dec", "b_enc", "b_dec"] + self.extra
        W = {"W_encT": self.W_encT, **{n: self.params[n] for n in names[1:]}}
        P = {n: W[n].numpy() for n in names}
        g = {n: (self._g[n] * coef).numpy().copy() for n in names}
        m = {n: self._m[n].numpy() for n in names}
        v = {n: self._v[n].numpy() for n in names}
        par = (g["W_dec"] * P["W_dec"]).sum(axis=1, keepdims=True)
        g["W_dec"] -= par * P["W_dec"]
        O.adam_step(P, g, m, v, lr, self.adam_step)
        self.params["W_enc"].copy_(self.W_encT.t())
