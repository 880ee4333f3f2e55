"""Host driver of the native SAE train step (pv_sae_* in include/pv_native.h).

All state lives in caller-owned torch tensors (plumbing): fp32 master parameters are the SAE
module's own ``nn.Parameter`` storage, gradients sit in ONE flat buffer
``[gW_enc^T | gW_dec | gb_enc | gb_dec]`` (feature-indexed rows first: a rank's optimizer shard is a
contiguous row range of each segment), Adam moments in two flat buffers of the same layout.

``W_enc`` additionally lives as ``W_encT`` (its fp32 transpose -- the layout the sparse backward
writes gradients in, Adam runs in and the exact re-scoring gathers rows from) and ``W_enc16T``
(fp16, the B operand of the filter GEMM); ``pv_sae_apply`` keeps all of them in step.  An in-place edit
of the parameter from outside (``load_state_dict``, ``optimizer.step()``, ``sae.W_enc.copy_(...)`` under
``no_grad``) is detected through the tensor version counter and answered with ``pv_sae_sync_shadows``.
Edits through ``param.data`` (``p.data.copy_``, ``p.data /= x``) do NOT move the version counter and are
invisible: after one, call ``engine.invalidate()`` (the package's own writers -- ``set_decoder_norm_to_unit_norm``,
the trainer's parameter gathers -- edit the Parameter itself, and ``VisionSAETrainer.checkpoint`` invalidates).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from .. import _native as N


def norm_mode_code(mode) -> int:
    """pv_sae_desc.normalize_layer_norm of cfg.normalize_activations (or of the bool the round 1-5 callers pass): 0 none, 1 "layer_norm"
    (sae.py:74-90), 2 "constant_norm_rescale" (sae.py:60-72)."""
    if mode is True or mode == "layer_norm" or mode == 1:
        return 1
    if mode == "constant_norm_rescale" or mode == 2:
        return 2
    if mode in (False, None, "none", 0):
        return 0
    raise ValueError(f"normalize_activations = {mode!r}")


class NativeSAE:
    def __init__(self, W_enc: torch.Tensor, W_dec: torch.Tensor, b_enc: torch.Tensor, b_dec: torch.Tensor, k: int,
                 layer_norm: bool, max_tokens: int, ln_eps: float = 1e-5, inference: bool = False,
                 b_dec_out: Optional[torch.Tensor] = None, W_skip: Optional[torch.Tensor] = None,
                 gated: Optional[Dict[str, torch.Tensor]] = None, gated_topk: bool = False,
                 tc_widths: Optional[Tuple[int, int]] = None, activation: str = "relu", lp_norm: float = 1.0):
        """inference=True: no gradient / Adam buffers (453 MB at 768 -> 24576): only ``encode_topk`` / ``forward``.
        b_dec_out [d_in] (+ W_skip [d_in, d_in]): a Transcoder (sae/transcoder.py; pv_sae_transcoder) -- ``step`` /
        ``dense_step`` then take the target activation, ``b_dec`` only centres the encoder input.
        gated = {b_gate, r_mag, b_mag} [d_sae] each: a GatedSparseAutoencoder (sae.py:648-792; pv_sae_gated) -- ``gated_step`` is
        its train step (``b_enc`` is kept but plays no part); gated_topk: its top-k form (activation_fn_str = "topk": TopK on the
        magnitudes and on the gate activations, k of each per token) -- ``gated_topk_step`` is the train step then, and the plan is
        created for twice the tokens (its k-dependent buffers hold both lists).
        tc_widths = (d_in, d_out) of a skip-less Transcoder between hook points of DIFFERENT width: every tensor given here is padded
        with zeros to D = max(d_in, d_out) (W_enc [D, d_sae], W_dec [d_sae, D], b_dec / b_dec_out [D]; pv_sae_transcoder.d_in_true /
        d_out_true); ``step`` / ``dense_step`` / ``relu_step`` take x [N, d_in] and target [N, d_out] and pad them into engine-owned
        buffers, the padding of the parameters stays exactly zero (its gradients are).
        layer_norm: cfg.normalize_activations ("layer_norm" / "constant_norm_rescale" / "none") or a bool (layer norm on / off).
        activation / lp_norm (the dense ReLU + L1 step only): "relu" or "tanh-relu" (sae.py:823-830), p of the sparsity term ||f_n||_p
        (sae.py:617; 1 or p > 1) -- anything but ("relu", 1) keeps ``relu_step`` on the dense GEMMs."""
        self.transcoder = b_dec_out is not None
        self.gated = gated is not None
        self.gated_topk = bool(gated_topk) and self.gated
        self.tc_widths = tuple(int(v) for v in tc_widths) if tc_widths is not None else None
        assert self.tc_widths is None or (self.transcoder and W_skip is None), "tc_widths: a transcoder without the skip connection"
        assert not (self.gated and (self.transcoder or inference)), "gated: training engine, no transcoder"
        assert W_skip is None or self.transcoder, "W_skip belongs to a transcoder (pass b_dec_out)"
        assert not (self.transcoder and inference), "the inference entry points do not serve a transcoder"
        for t in (W_enc, W_dec, b_enc, b_dec) + tuple(t for t in (b_dec_out, W_skip) if t is not None) + tuple((gated or {}).values()):
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise N.NativeError("native SAE needs contiguous fp32 CUDA parameters")
        self.lib = N.lib()
        self.device = W_enc.device
        self.d_in, self.d_sae = W_enc.shape
        assert tuple(W_dec.shape) == (self.d_sae, self.d_in)
        self.k = int(k)
        self.max_tokens = int(max_tokens)
        # the tensors whose version counters tell about outside edits (nn.Parameters when the trainer passes them)
        self._src = dict(W_enc=W_enc, W_dec=W_dec, b_enc=b_enc, b_dec=b_dec)
        if self.transcoder:
            assert tuple(b_dec_out.shape) == (self.d_in,) and (W_skip is None or tuple(W_skip.shape) == (self.d_in, self.d_in))
            self._src["b_dec_out"] = b_dec_out
            if W_skip is not None:
                self._src["W_skip"] = W_skip
        if self.gated:
            assert sorted(gated) == ["b_gate", "b_mag", "r_mag"] and all(tuple(t.shape) == (self.d_sae,) for t in gated.values())
            self._src.update(gated)
        self.params = {n: t.detach() for n, t in self._src.items()}
        desc = N.SaeDesc(d_in=self.d_in, d_sae=self.d_sae, k=self.k, normalize_layer_norm=norm_mode_code(layer_norm),
                         max_tokens=self.max_tokens * (2 if self.gated_topk else 1), ln_eps=ln_eps,
                         activation={"relu": 0, "topk": 0, "tanh-relu": 1}[activation], lp_norm=float(lp_norm))
        self._plan = C.c_void_p()
        N.check(self.lib.pv_sae_plan_create(C.byref(desc), C.byref(self._plan)), "pv_sae_plan_create")
        self.filtered_encoder = bool(self.lib.pv_sae_encoder_is_filtered(self._plan))
        dev = self.device
        nW = self.d_in * self.d_sae
        self.n_flat = 2 * nW + self.d_sae + self.d_in
        if self.transcoder:
            self.n_flat += self.d_in + (self.d_in * self.d_in if W_skip is not None else 0)
        if self.gated:
            self.n_flat += 3 * self.d_sae
        f32 = dict(dtype=torch.float32, device=dev)
        self.inference = bool(inference)
        n_alloc = 0 if inference else self.n_flat
        self.flat_g = torch.zeros(n_alloc, **f32)
        self.flat_m = torch.zeros(n_alloc, **f32)
        self.flat_v = torch.zeros(n_alloc, **f32)

        def views(flat: torch.Tensor) -> Dict[str, torch.Tensor]:
            o = 0
            out = {}
            if flat.numel() == 0:                                  # inference engine: null pointers
                z = flat.view(0, self.d_in)
                return dict(W_encT=z, W_dec=z, b_enc=flat, b_dec=flat)
            out["W_encT"] = flat[o:o + nW].view(self.d_sae, self.d_in); o += nW
            out["W_dec"] = flat[o:o + nW].view(self.d_sae, self.d_in); o += nW
            out["b_enc"] = flat[o:o + self.d_sae]; o += self.d_sae
            out["b_dec"] = flat[o:o + self.d_in]; o += self.d_in
            if self.transcoder:
                out["b_dec_out"] = flat[o:o + self.d_in]; o += self.d_in
                if "W_skip" in self._src:
                    out["W_skip"] = flat[o:o + self.d_in * self.d_in].view(self.d_in, self.d_in)
            if self.gated:
                for name in ("b_gate", "r_mag", "b_mag"):
                    out[name] = flat[o:o + self.d_sae]; o += self.d_sae
            return out

        def param_layout(v: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
            out = dict(W_enc=v["W_encT"].t(), W_dec=v["W_dec"], b_enc=v["b_enc"], b_dec=v["b_dec"])
            out.update({n: v[n] for n in ("b_dec_out", "W_skip", "b_gate", "r_mag", "b_mag") if n in v})
            return out

        self._g, self._m, self._v = views(self.flat_g), views(self.flat_m), views(self.flat_v)
        # views in the parameters' own layouts (W_enc: a transposed, non-contiguous view)
        self.g = dict(W_enc=self._g["W_encT"], W_dec=self._g["W_dec"], b_enc=self._g["b_enc"], b_dec=self._g["b_dec"])   # NB: g["W_enc"] is TRANSPOSED
        self.g.update({n: self._g[n] for n in ("b_dec_out", "W_skip", "b_gate", "r_mag", "b_mag") if n in self._g})
        self.m, self.v = param_layout(self._m), param_layout(self._v)
        # encoder shadows
        self.W_encT = torch.empty(self.d_sae, self.d_in, **f32)
        self.W_enc16T = torch.empty(self.d_sae, self.d_in, dtype=torch.float16, device=dev)
        self.enc_colsq = torch.zeros(self.d_sae, **f32)
        self.dec_inv_norm = torch.ones(self.d_sae, **f32)
        self.act_freq_scores = torch.zeros(self.d_sae, **f32)
        self.n_fwd_since_fired = torch.zeros(self.d_sae, **f32)
        self.fire_count = torch.zeros(self.d_sae, **f32)
        self.scalars = torch.zeros(8, **f32)
        self.sq_partial = torch.zeros(1024, **f32)
        rows = self.max_tokens * (2 if self.gated_topk else 1)             # (top-k gated: the magnitude list, then the gate list)
        self.topk_idx = torch.zeros(rows, self.k, dtype=torch.int32, device=dev)
        self.topk_val = torch.zeros(rows, self.k, **f32)
        self.sae_out = torch.zeros(self.max_tokens, self.d_in, **f32)
        self.workspace = torch.empty(self.lib.pv_sae_workspace_bytes(self._plan), dtype=torch.uint8, device=dev)
        self._tc_scratch = None
        self._target: Optional[torch.Tensor] = None                # transcoder: the target of the coming / last step
        if self.transcoder and W_skip is not None:
            self._tc_scratch = torch.empty(self.lib.pv_sae_transcoder_scratch_bytes(self._plan, self.max_tokens), dtype=torch.uint8,
                                           device=dev)
        self._gt_scratch = None
        self._gk_scratch = None
        if self.gated_topk:
            self._gk_scratch = torch.empty(self.lib.pv_sae_gated_topk_scratch_bytes(self._plan, self.max_tokens), dtype=torch.uint8,
                                           device=dev)
        elif self.gated:
            self._gt_scratch = torch.empty(self.lib.pv_sae_gated_scratch_bytes(self._plan, self.max_tokens), dtype=torch.uint8, device=dev)
        self.adam_step = 0
        self.relu_cap = 256                                        # per-token capacity of relu_step's sparse form
        self._relu_ws: Optional[torch.Tensor] = None
        self._shadow_key: Optional[Tuple[int, int]] = None
        self._inv_norm_key: Optional[Tuple[int, int]] = None     # W_dec as the last full-range apply left it (dec_inv_norm is current)
        self._grad_fresh = False                                   # gradient buffers exactly as the last step wrote them
        self._sq_fused = False
        # lazy_w_enc: ``apply`` keeps the encoder in W_encT (+ fp16 shadow) only and leaves the parameter's own [d_in, d_sae]
        # layout stale (its transposed write is 75 MB per step at 768 -> 24576); ``materialize_w_enc()`` rewrites it.  The
        # kernels never read that layout.  Off by default: whoever turns it on (VisionSAETrainer, single process) also makes
        # sure readers of the parameter trigger the materialisation (StandardSparseAutoencoder._native_sync).
        self.lazy_w_enc = False
        self._w_enc_stale = False
        self.sync_shadows()

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None and self._plan.value:
                self.lib.pv_sae_plan_destroy(self._plan)
                self._plan = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _state(self, with_w_enc: bool = False) -> N.SaeState:
        P, g, m, v = self.params, self._g, self._m, self._v
        if self.inference:
            return N.SaeState(W_enc=P["W_enc"].data_ptr(), W_dec=P["W_dec"].data_ptr(), b_enc=P["b_enc"].data_ptr(),
                              b_dec=P["b_dec"].data_ptr(), W_encT=self.W_encT.data_ptr(), W_enc16T=self.W_enc16T.data_ptr(),
                              enc_colsq=self.enc_colsq.data_ptr())
        return N.SaeState(
            W_enc=None if (self.lazy_w_enc and not with_w_enc) else P["W_enc"].data_ptr(),
            W_dec=P["W_dec"].data_ptr(), b_enc=P["b_enc"].data_ptr(), b_dec=P["b_dec"].data_ptr(),
            gW_enc=g["W_encT"].data_ptr(), gW_dec=g["W_dec"].data_ptr(), gb_enc=g["b_enc"].data_ptr(), gb_dec=g["b_dec"].data_ptr(),
            mW_enc=m["W_encT"].data_ptr(), mW_dec=m["W_dec"].data_ptr(), mb_enc=m["b_enc"].data_ptr(), mb_dec=m["b_dec"].data_ptr(),
            vW_enc=v["W_encT"].data_ptr(), vW_dec=v["W_dec"].data_ptr(), vb_enc=v["b_enc"].data_ptr(), vb_dec=v["b_dec"].data_ptr(),
            act_freq_scores=self.act_freq_scores.data_ptr(), n_fwd_since_fired=self.n_fwd_since_fired.data_ptr(),
            W_encT=self.W_encT.data_ptr(), W_enc16T=self.W_enc16T.data_ptr(), enc_colsq=self.enc_colsq.data_ptr(),
            dec_inv_norm=self.dec_inv_norm.data_ptr(), tc=self._tc_state(), gt=self._gt_state())

    def _gt_state(self) -> N.SaeGated:
        if not self.gated:
            return N.SaeGated()
        P, g, m, v = self.params, self._g, self._m, self._v
        kw = {}
        for name, short in (("b_gate", "b_gate"), ("r_mag", "r_mag"), ("b_mag", "b_mag")):
            kw[short] = P[name].data_ptr()
            kw["g" + short] = g[name].data_ptr()
            kw["m" + short] = m[name].data_ptr()
            kw["v" + short] = v[name].data_ptr()
        sc = self._gt_scratch
        return N.SaeGated(scratch=sc.data_ptr() if sc is not None else None, scratch_bytes=sc.numel() if sc is not None else 0, **kw)

    def _tc_state(self) -> N.SaeTranscoder:
        if not self.transcoder:
            return N.SaeTranscoder()
        P, g, m, v = self.params, self._g, self._m, self._v
        skip = "W_skip" in P
        sc = self._tc_scratch
        return N.SaeTranscoder(
            b_dec_out=P["b_dec_out"].data_ptr(), gb_dec_out=g["b_dec_out"].data_ptr(), mb_dec_out=m["b_dec_out"].data_ptr(),
            vb_dec_out=v["b_dec_out"].data_ptr(),
            W_skip=P["W_skip"].data_ptr() if skip else None, gW_skip=g["W_skip"].data_ptr() if skip else None,
            mW_skip=m["W_skip"].data_ptr() if skip else None, vW_skip=v["W_skip"].data_ptr() if skip else None,
            target=self._target.data_ptr() if self._target is not None else None,
            scratch=sc.data_ptr() if sc is not None else None, scratch_bytes=sc.numel() if sc is not None else 0,
            d_in_true=self.tc_widths[0] if self.tc_widths else 0, d_out_true=self.tc_widths[1] if self.tc_widths else 0)

    def _set_target(self, x: torch.Tensor, target: Optional[torch.Tensor]) -> None:
        if not self.transcoder:
            assert target is None, "a target is a transcoder's business"
            return
        if target is None:
            raise ValueError("transcoder step: the target activation is required")
        t = target.to(torch.float32).contiguous()
        if self.tc_widths is not None:
            # rows padded to the common width (the step writes the padding columns itself: pv_sae_transcoder.d_out_true)
            if tuple(t.shape) != (x.shape[0], self.tc_widths[1]) or t.device != self.device:
                raise ValueError(f"target {tuple(t.shape)} on {t.device}: expected [{x.shape[0]}, {self.tc_widths[1]}] on {self.device}")
            if getattr(self, "_y_pad", None) is None:
                self._y_pad = torch.zeros(self.max_tokens, self.d_in, dtype=torch.float32, device=self.device)
            buf = self._y_pad[:x.shape[0]]
            buf[:, :self.tc_widths[1]].copy_(t)
            self._target = buf
            return
        if tuple(t.shape) != tuple(x.shape) or t.device != self.device:
            raise ValueError(f"target {tuple(t.shape)} on {t.device} does not match the input {tuple(x.shape)} on {self.device}")
        self._target = t

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _bm(self, batch_mean: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """The caller's global batch mean as the kernels read it: d_in floats (a transcoder of unequal widths: the TARGET's mean,
        d_out floats, padded to the common width)."""
        if batch_mean is None:
            return None
        bm = batch_mean.to(torch.float32).contiguous().view(-1)
        if self.tc_widths is not None and bm.numel() == self.tc_widths[1] and bm.numel() < self.d_in:
            bm = torch.nn.functional.pad(bm, (0, self.d_in - bm.numel()))
        if bm.numel() != self.d_in:
            raise ValueError(f"batch_mean has {bm.numel()} entries, expected {self.d_in}")
        return bm

    def _check_x(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        if self.tc_widths is not None and x.ndim == 2 and x.shape[1] == self.tc_widths[0] and self.tc_widths[0] < self.d_in:
            if x.shape[0] > self.max_tokens or x.device != self.device:
                raise ValueError(f"expected [N<={self.max_tokens}, {self.tc_widths[0]}] on {self.device}, got {tuple(x.shape)} on {x.device}")
            if getattr(self, "_x_pad", None) is None:
                self._x_pad = torch.zeros(self.max_tokens, self.d_in, dtype=torch.float32, device=self.device)
            buf = self._x_pad[:x.shape[0]]
            buf[:, :self.tc_widths[0]].copy_(x)                     # (the padding columns stay zero)
            return buf
        if x.ndim != 2 or x.shape[1] != self.d_in or x.shape[0] > self.max_tokens or x.device != self.device:
            raise ValueError(f"expected [N<={self.max_tokens}, {self.d_in}] on {self.device}, got {tuple(x.shape)} on {x.device}")
        return x

    # ---- encoder shadows -------------------------------------------------------------------------
    def _w_enc_key(self) -> Tuple[int, int]:
        t = self._src["W_enc"]
        return (t.data_ptr(), t._version)

    def sync_shadows(self, from_transposed: bool = False, j_lo: int = 0, j_hi: Optional[int] = None) -> None:
        """Rebuild W_encT / W_enc16T / enc_colsq from W_enc (default), or W_enc / W_enc16T / enc_colsq from W_encT."""
        st = self._state(with_w_enc=True)
        N.check(self.lib.pv_sae_sync_shadows(self._plan, C.byref(st), int(from_transposed), int(j_lo),
                                             int(self.d_sae if j_hi is None else j_hi), self._stream()), "pv_sae_sync_shadows")
        self._shadow_key = self._w_enc_key()
        if j_lo == 0 and (j_hi is None or j_hi == self.d_sae):
            self._w_enc_stale = False                              # both layouts agree again

    def materialize_w_enc(self) -> None:
        """lazy_w_enc: bring the parameter's own layout up to date with the transposed master the kernels train (no-op when
        it is)."""
        if self._w_enc_stale:
            self.sync_shadows(from_transposed=True)

    def invalidate(self) -> None:
        """Forget everything derived from the parameters (encoder shadows, the decoder's inverse row norms): the next
        call rebuilds it.  For edits the version counters cannot show (``param.data`` writes, raw-pointer writers)."""
        self._shadow_key = None
        self._inv_norm_key = None

    def _ensure_shadows(self) -> None:
        """An in-place edit of W_enc from outside (optimizer.step() of another trainer, load_state_dict, .copy_) bumps
        the tensor's version counter; the library's own updates go through raw pointers and do not."""
        if self._shadow_key != self._w_enc_key():
            self.sync_shadows()                                    # (the outside edit is the truth now, stale or not)

    # ---- the step ----------------------------------------------------------------------------------
    def _w_dec_key(self) -> Tuple[int, int]:
        t = self._src["W_dec"]
        return (t.data_ptr(), t._version)

    def renorm_decoder(self) -> None:
        self._inv_norm_key = None
        st = self._state()
        N.check(self.lib.pv_sae_renorm_decoder(self._plan, C.byref(st), self._stream()), "pv_sae_renorm_decoder")

    def step(self, x: torch.Tensor, batch_mean: Optional[torch.Tensor] = None, n_global: Optional[int] = None,
             update_stats: bool = True, want_out: bool = False, renorm_decoder: bool = False,
             sparse_grads: bool = False, target: Optional[torch.Tensor] = None, fused_sqnorm: bool = False) -> None:
        """forward + backward + statistics; gradients are written into ``flat_g``; scalars[0..2] =
        loss, mse_loss, l0 (device).  renorm_decoder: set_decoder_norm_to_unit_norm as part of the step (the rewrite of
        W_dec is fused into the following ``apply``) instead of a separate ``renorm_decoder()`` pass.  sparse_grads
        (PV_SAE_SPARSE_GRADS, single process): the gradient rows of features that kept no token are left unwritten and
        ``apply`` takes them as zero -- ``flat_g`` is then NOT a complete gradient and only ``grad_sqnorm(from_step=True)``
        and ``apply`` may follow.  target (transcoder engines): the activation to reconstruct; batch_mean is then ITS mean.
        fused_sqnorm (PV_SAE_FUSED_SQNORM; ignored on transcoders, whose clip norm has more terms): the step's last launch also
        leaves scalars[3], the clip norm's sum of squares -- a following ``grad_sqnorm(from_step=True)`` then has nothing to do."""
        x = self._check_x(x)
        self._set_target(x, target)
        self._ensure_shadows()
        n = x.shape[0]
        st = self._state()
        # the last full-range apply left 1 / ||W_dec[j]|| of the rows it wrote; good as long as nobody touched W_dec since
        inv_valid = renorm_decoder and self._inv_norm_key is not None and self._inv_norm_key == self._w_dec_key()
        out = N.SaeOut(sae_out=self.sae_out.data_ptr() if want_out else None, topk_idx=self.topk_idx.data_ptr(),
                       topk_val=self.topk_val.data_ptr(), scalars=self.scalars.data_ptr(),
                       fire_count=self.fire_count.data_ptr())
        bm = None
        if batch_mean is not None:
            bm = self._bm(batch_mean)
        fuse = bool(fused_sqnorm) and not self.transcoder and self.d_in <= 4096
        N.check(self.lib.pv_sae_step(self._plan, C.byref(st), x.data_ptr(), n, bm.data_ptr() if bm is not None else None,
                                     int(n_global if n_global is not None else n),
                                     int(bool(update_stats)) | (2 if renorm_decoder else 0) | (4 if inv_valid else 0) |
                                     (8 if sparse_grads else 0) | (32 if fuse else 0), C.byref(out),
                                     self.workspace.data_ptr(), self.workspace.numel(), self._stream()), "pv_sae_step")
        self._grad_fresh = True
        self._grad_sparse = bool(sparse_grads)
        self._sq_fused = fuse

    def dense_step(self, x: torch.Tensor, l1_coefficient: float, batch_mean: Optional[torch.Tensor] = None,
                   n_global: Optional[int] = None, update_stats: bool = True, want_out: bool = False,
                   renorm_decoder: bool = True, dead_mask: Optional[torch.Tensor] = None,
                   target: Optional[torch.Tensor] = None, ghost_global=None) -> None:
        """The ReLU + L1 step (pv_sae_dense_step): forward + backward + statistics on dense fp32 MFMA GEMMs with fused
        epilogues; gradients are written into ``flat_g`` (complete); scalars = loss, mse_loss, l0, -, l1_loss, ghost loss.  The
        engine's ``k`` plays no role.  dead_mask [d_sae] bool (use_ghost_grads: ``n_forward_passes_since_fired >
        dead_feature_window`` BEFORE this step, train_sae.py:330-332): adds the ghost residual loss and its gradient
        (sae.py:151-179); costs one device read-back (the number of dead features sizes three small GEMMs).  target: as in ``step``.
        ghost_global: as in ``_ghost_struct`` (required for ghost gradients with n_global > the tokens of this call)."""
        x = self._check_x(x)
        self._set_target(x, target)
        self._ensure_shadows()                                    # (the encoder is read as W_encT: an outside edit of W_enc must reach it)
        n = x.shape[0]
        st = self._state()
        out = N.SaeOut(sae_out=self.sae_out.data_ptr() if want_out else None, topk_idx=None, topk_val=None,
                       scalars=self.scalars.data_ptr(), fire_count=self.fire_count.data_ptr())
        bm = self._bm(batch_mean)
        ghost = self._ghost_struct(dead_mask, n, ghost_global) if dead_mask is not None else None
        N.check(self.lib.pv_sae_dense_step(self._plan, C.byref(st), x.data_ptr(), n, bm.data_ptr() if bm is not None else None,
                                           int(n_global if n_global is not None else n),
                                           int(bool(update_stats)) | (2 if renorm_decoder else 0), float(l1_coefficient),
                                           C.byref(ghost) if ghost is not None else None,
                                           C.byref(out), self.workspace.data_ptr(), self.workspace.numel(), self._stream()),
                "pv_sae_dense_step")
        self._inv_norm_key = None
        self._grad_fresh = False
        self._sq_fused = False
        self._grad_sparse = False

    def relu_step(self, x: torch.Tensor, l1_coefficient: float, batch_mean: Optional[torch.Tensor] = None,
                  n_global: Optional[int] = None, update_stats: bool = True, want_out: bool = False,
                  renorm_decoder: bool = True, target: Optional[torch.Tensor] = None, cap: Optional[int] = None,
                  sparse_grads: bool = False) -> None:
        """The ReLU + L1 step, sparse where the batch allows it (pv_sae_relu_step): ONE fp16-filtered product over all features +
        exact fp32 re-scoring gives every token's positive activations as a list of at most ``cap`` pairs and the k-sparse kernels do
        the rest; a batch some token of which holds more than that runs on the dense GEMMs of ``dense_step`` instead -- decided on the
        GPU (``relu_mode``: 0 sparse, 1 dense; a device word), same results either way.  Contract, scalars and follow-up calls as
        ``dense_step`` (no ghost gradients: those stay with ``dense_step``).  renorm_decoder is deferred as in ``step`` (inverse norms
        now, rows rewritten by ``apply``; a step that goes dense rewrites them itself).  sparse_grads (single process): as in ``step``
        -- only ``grad_sqnorm(from_step=True)`` and ``apply`` may follow (a step that ran dense marks every feature live)."""
        x = self._check_x(x)
        self._set_target(x, target)
        self._ensure_shadows()
        n = x.shape[0]
        cap = int(cap if cap is not None else self.relu_cap)
        key = (self.max_tokens, cap)
        if getattr(self, "_relu_key", None) != key:
            need = self.lib.pv_sae_relu_workspace_bytes(self._plan, self.max_tokens, cap)
            if not need:
                raise ValueError(f"relu_step: cap must be a multiple of 4 in [4, 256], got {cap}")
            self._relu_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
            self._relu_key = key
        st = self._state()
        out = N.SaeOut(sae_out=self.sae_out.data_ptr() if want_out else None, topk_idx=None, topk_val=None,
                       scalars=self.scalars.data_ptr(), fire_count=self.fire_count.data_ptr())
        bm = self._bm(batch_mean)
        sp = N.SaeReluSparse(cap=cap, reserved=0, workspace=self._relu_ws.data_ptr(), workspace_bytes=self._relu_ws.numel())
        self._relu_last = (n, cap)
        inv_valid = renorm_decoder and self._inv_norm_key is not None and self._inv_norm_key == self._w_dec_key()
        N.check(self.lib.pv_sae_relu_step(self._plan, C.byref(st), x.data_ptr(), n, bm.data_ptr() if bm is not None else None,
                                          int(n_global if n_global is not None else n),
                                          int(bool(update_stats)) | (2 if renorm_decoder else 0) | (4 if inv_valid else 0) |
                                          (8 if sparse_grads else 0), float(l1_coefficient), C.byref(sp),
                                          C.byref(out), self.workspace.data_ptr(), self.workspace.numel(), self._stream()),
                "pv_sae_relu_step")
        self._inv_norm_key = None
        self._grad_fresh = bool(sparse_grads)                     # (the per-feature clip-norm terms exist in that form only)
        self._sq_fused = False
        self._grad_sparse = bool(sparse_grads)

    def _relu_region(self, name: bytes, dtype: torch.dtype, shape) -> torch.Tensor:
        n, cap = self._relu_last
        off = self.lib.pv_debug_sae_relu_offset(self._plan, n, cap, name)
        numel = 1
        for s_ in shape:
            numel *= s_
        return self._relu_ws[off:off + numel * 4].view(dtype).view(shape)

    @property
    def relu_mode(self) -> torch.Tensor:
        """Device word of the last ``relu_step``: 0 = it ran sparse, 1 = on the dense GEMMs."""
        return self._relu_region(b"mode", torch.int32, (1,))

    def relu_pairs(self):
        """(idx [N, cap] int32, val [N, cap], count [N] int32) of the last ``relu_step`` that ran sparse: token n keeps the features
        idx[n, :count[n]] with the activations val[n, :count[n]] (value descending); for tests."""
        n, cap = self._relu_last
        return (self._relu_region(b"idx", torch.int32, (n, cap)), self._relu_region(b"val", torch.float32, (n, cap)),
                self._relu_region(b"tok_cnt", torch.int32, (n,)))

    def _ghost_struct(self, dead_mask: torch.Tensor, n: int, ghost_global=None) -> "N.SaeGhost":
        """pv_sae_ghost for the features ``dead_mask`` marks (one device read-back: their number sizes three small GEMMs).
        ghost_global = (err_colmean [d_in], mse [1], n_global): the residual's column mean and the mse loss of the GLOBAL batch when
        the tokens are sharded over ranks (sae.py:156, :172 take them over the whole batch)."""
        idx = torch.nonzero(dead_mask.to(self.device), as_tuple=False).flatten().to(torch.int32)     # (synchronises: n_dead is a launch size)
        nd = int(idx.numel())
        slot = torch.full((self.d_sae,), -1, dtype=torch.int32, device=self.device)
        if nd:
            slot[idx.long()] = torch.arange(nd, dtype=torch.int32, device=self.device)
        need = self.lib.pv_sae_ghost_workspace_bytes(self._plan, n, nd)
        gws = getattr(self, "_ghost_ws", None)
        if gws is None or gws.numel() < need:
            self._ghost_ws = gws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ghost_keep = (idx, slot)                            # (alive until the kernels have run)
        extra = {}
        if ghost_global is not None:
            cm, mse, ng = ghost_global
            cm, mse = cm.to(torch.float32).contiguous(), mse.to(torch.float32).contiguous().view(-1)
            assert cm.numel() == self.d_in and mse.numel() == 1 and cm.is_cuda and mse.is_cuda
            self._ghost_keep += (cm, mse)
            extra = dict(err_colmean=cm.data_ptr(), mse_global=mse.data_ptr(), n_global=int(ng))
        return N.SaeGhost(n_dead=nd, dead_idx=idx.data_ptr() if nd else None, dead_slot=slot.data_ptr(),
                          workspace=gws.data_ptr(), workspace_bytes=gws.numel(), **extra)

    def topk_ghost(self, x: torch.Tensor, dead_mask: torch.Tensor, ghost_global=None) -> None:
        """Ghost gradients of a top-k SAE (pv_sae_topk_ghost; sae.py:151-179 with train_sae.py:330-346), added to what the
        preceding ``step(x, renorm_decoder=False, sparse_grads=False, want_out=True)`` left (the decoder renormalised by
        ``renorm_decoder()`` before it): scalars[5] = ghost residual loss, scalars[0] = mse + ghost, the dead features' gradient rows
        and gb_dec updated.  dead_mask [d_sae] bool = ``n_forward_passes_since_fired > dead_feature_window`` BEFORE that step.
        ghost_global: as in ``_ghost_struct`` (tokens sharded over ranks: the preceding step ran with batch_mean / n_global).
        Follow with ``grad_sqnorm()`` (the full pass) and ``apply``."""
        x = self._check_x(x)
        n = x.shape[0]
        assert not self.gated and not (self.transcoder and (self.tc_widths is not None or ghost_global is not None))
        st = self._state()
        out = N.SaeOut(sae_out=self.sae_out.data_ptr(), topk_idx=None, topk_val=None, scalars=self.scalars.data_ptr(),
                       fire_count=self.fire_count.data_ptr())
        ghost = self._ghost_struct(dead_mask, n, ghost_global)
        N.check(self.lib.pv_sae_topk_ghost(self._plan, C.byref(st), x.data_ptr(), n, C.byref(ghost), C.byref(out),
                                           self.workspace.data_ptr(), self.workspace.numel(), self._stream()), "pv_sae_topk_ghost")
        self._grad_fresh = False                                  # (the per-feature norm terms of the step no longer describe the buffers)
        self._sq_fused = False
        self._grad_sparse = False

    def gated_step(self, x: torch.Tensor, l1_coefficient: float, batch_mean: Optional[torch.Tensor] = None,
                   n_global: Optional[int] = None, update_stats: bool = True, want_out: bool = False,
                   cap: Optional[int] = None, sparse: bool = True) -> None:
        """One train step of a gated SAE: set_decoder_norm_to_unit_norm, forward, backward, statistics;
        gradients in ``flat_g`` (complete); scalars = loss, mse_loss, l0, -, l1_loss, -, auxiliary loss.  batch_mean / n_global:
        as in ``step`` (tokens sharded over ranks; the caller all-reduces ``flat_g``).  sparse (default; pv_sae_gated_step_sparse):
        every token's OPEN gates as a list of at most ``cap`` pairs and the k-sparse kernels behind them, the dense GEMMs
        (pv_sae_gated_step) when a token of the batch holds more -- decided on the GPU (``gated_mode``: 0 sparse, 1 dense), same
        results either way; sparse=False: the dense GEMMs unconditionally."""
        assert self.gated
        x = self._check_x(x)
        self._ensure_shadows()                                    # (as in step / dense_step)
        n = x.shape[0]
        st = self._state()
        out = N.SaeOut(sae_out=self.sae_out.data_ptr() if want_out else None, topk_idx=None, topk_val=None,
                       scalars=self.scalars.data_ptr(), fire_count=self.fire_count.data_ptr())
        bm = self._bm(batch_mean)
        flags = int(bool(update_stats)) | 2
        ng = int(n_global if n_global is not None else n)
        if sparse:
            cap = int(cap if cap is not None else self.relu_cap)
            key = (self.max_tokens, cap)
            if getattr(self, "_gated_key", None) != key:
                need = self.lib.pv_sae_gated_sparse_workspace_bytes(self._plan, self.max_tokens, cap)
                if not need:
                    raise ValueError(f"gated_step: cap must be a multiple of 4 in [4, 256], got {cap}")
                self._gated_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
                self._gated_key = key
            sp = N.SaeReluSparse(cap=cap, reserved=0, workspace=self._gated_ws.data_ptr(), workspace_bytes=self._gated_ws.numel())
            N.check(self.lib.pv_sae_gated_step_sparse(self._plan, C.byref(st), x.data_ptr(), n, bm.data_ptr() if bm is not None else None,
                                                      ng, flags, float(l1_coefficient), C.byref(sp), C.byref(out),
                                                      self.workspace.data_ptr(), self.workspace.numel(), self._stream()),
                    "pv_sae_gated_step_sparse")
        else:
            N.check(self.lib.pv_sae_gated_step(self._plan, C.byref(st), x.data_ptr(), n, bm.data_ptr() if bm is not None else None,
                                               ng, flags, float(l1_coefficient), C.byref(out), self.workspace.data_ptr(),
                                               self.workspace.numel(), self._stream()), "pv_sae_gated_step")
        self._inv_norm_key = None
        self._grad_fresh = False
        self._sq_fused = False
        self._grad_sparse = False

    def gated_topk_step(self, x: torch.Tensor, batch_mean: Optional[torch.Tensor] = None, n_global: Optional[int] = None,
                        update_stats: bool = True, want_out: bool = False) -> None:
        """One train step of a gated SAE in its top-k form (pv_sae_gated_topk_step): contract as ``gated_step`` (scalars[4] = 0: no L1
        term).  ``topk_idx / topk_val`` rows [0, n) = the magnitude list (feature_acts), rows [n, 2 n) = the gate list."""
        assert self.gated_topk
        x = self._check_x(x)
        self._ensure_shadows()
        n = x.shape[0]
        st = self._state()
        out = N.SaeOut(sae_out=self.sae_out.data_ptr() if want_out else None, topk_idx=self.topk_idx.data_ptr(),
                       topk_val=self.topk_val.data_ptr(), scalars=self.scalars.data_ptr(), fire_count=self.fire_count.data_ptr())
        bm = self._bm(batch_mean)
        N.check(self.lib.pv_sae_gated_topk_step(self._plan, C.byref(st), x.data_ptr(), n, bm.data_ptr() if bm is not None else None,
                                                int(n_global if n_global is not None else n), int(bool(update_stats)) | 2, C.byref(out),
                                                self.workspace.data_ptr(), self.workspace.numel(), self._gk_scratch.data_ptr(),
                                                self._gk_scratch.numel(), self._stream()), "pv_sae_gated_topk_step")
        self._inv_norm_key = None
        self._grad_fresh = False
        self._sq_fused = False
        self._grad_sparse = False

    @property
    def gated_mode(self) -> int:
        """0: the last ``gated_step(sparse=True)`` ran sparse; 1: it ran on the dense GEMMs (a device word: this read synchronises)."""
        return int(self._gated_ws[:4].view(torch.int32)[0].item())

    def _no_pending_sparse(self, what: str) -> None:
        if getattr(self, "_grad_sparse", False) and self._grad_fresh:
            raise RuntimeError(f"{what} between a sparse_grads step and its apply(): the step's workspace (the list of live "
                               "features) and its partial gradient buffers are only good for grad_sqnorm(from_step=True) / apply()")

    def grad_sqnorm(self, from_step: bool = False) -> None:
        """scalars[3] = sum of squares of the whole gradient.  from_step: take it from the per-feature terms the backward
        kernels of the last ``step`` left behind instead of re-reading the 151 MB -- only while the gradient buffers are
        exactly what that step wrote (no all-reduce, no edit through ``g``); otherwise the full pass runs."""
        if from_step and self._grad_fresh:
            if getattr(self, "_sq_fused", False):                 # (the step's last launch left scalars[3]: PV_SAE_FUSED_SQNORM)
                return
            st = self._state()
            N.check(self.lib.pv_sae_grad_sqnorm_step(self._plan, C.byref(st), self.workspace.data_ptr(), self.scalars.data_ptr(),
                                                     self._stream()), "pv_sae_grad_sqnorm_step")
            return
        self._no_pending_sparse("a full pass over the gradient buffers")
        N.check(self.lib.pv_sae_grad_sqnorm(self.flat_g.data_ptr(), self.n_flat, self.sq_partial.data_ptr(),
                                            self.scalars.data_ptr(), self._stream()), "pv_sae_grad_sqnorm")

    def grad_sqnorm_rows(self, j_lo: int, j_hi: int, include_b_dec: bool) -> None:
        """scalars[3] = sum of squares of the gradient rows of features [j_lo, j_hi) (+ gb_dec): one rank's term."""
        self._no_pending_sparse("grad_sqnorm_rows")
        st = self._state()
        N.check(self.lib.pv_sae_grad_sqnorm_rows(self._plan, C.byref(st), int(j_lo), int(j_hi), int(include_b_dec),
                                                 self.sq_partial.data_ptr(), self.scalars.data_ptr(), self._stream()),
                "pv_sae_grad_sqnorm_rows")

    def apply(self, lr: float, max_grad_norm: Optional[float], j_lo: int = 0, j_hi: Optional[int] = None) -> None:
        """clip -> project -> Adam for the features [j_lo, j_hi) (default: all) and b_dec."""
        self.adam_step += 1
        st = self._state()
        N.check(self.lib.pv_sae_apply(self._plan, C.byref(st), self.scalars.data_ptr(),
                                      float(max_grad_norm) if max_grad_norm else -1.0, float(lr), self.adam_step,
                                      int(j_lo), int(self.d_sae if j_hi is None else j_hi), self._stream()), "pv_sae_apply")
        self._grad_fresh = False
        self._sq_fused = False
        full = j_lo == 0 and (j_hi is None or j_hi == self.d_sae)
        self._inv_norm_key = self._w_dec_key() if full else None
        if self.lazy_w_enc:
            self._w_enc_stale = True

    def encode_topk(self, x: torch.Tensor, want_ln_stats: bool = True):
        """(idx [N,k] int32, val [N,k], mu [N], std [N]) -- the sparse form of feature_acts.  want_ln_stats=False: no copies
        of the LayerNorm statistics out of the workspace (mu, std are None)."""
        self._no_pending_sparse("encode_topk")
        x = self._check_x(x)
        self._ensure_shadows()
        n = x.shape[0]
        st = self._state()
        mu = sd = None
        if want_ln_stats:
            mu = torch.empty(n, dtype=torch.float32, device=self.device)
            sd = torch.empty(n, dtype=torch.float32, device=self.device)
        N.check(self.lib.pv_sae_encode_topk(self._plan, C.byref(st), x.data_ptr(), n, self.topk_idx.data_ptr(),
                                            self.topk_val.data_ptr(), mu.data_ptr() if want_ln_stats else None,
                                            sd.data_ptr() if want_ln_stats else None,
                                            self.workspace.data_ptr(), self.workspace.numel(), self._stream()),
                "pv_sae_encode_topk")
        return self.topk_idx[:n], self.topk_val[:n], mu, sd

    def forward(self, x: torch.Tensor):
        """Inference: (sae_out [N, d_in], idx [N, k] int32, val [N, k]); the views are overwritten by the next call."""
        self._no_pending_sparse("forward")
        x = self._check_x(x)
        self._ensure_shadows()
        n = x.shape[0]
        st = self._state()
        N.check(self.lib.pv_sae_forward(self._plan, C.byref(st), x.data_ptr(), n, self.sae_out.data_ptr(), self.topk_idx.data_ptr(),
                                        self.topk_val.data_ptr(), None, None, self.scalars.data_ptr(), self.workspace.data_ptr(),
                                        self.workspace.numel(), self._stream()), "pv_sae_forward")
        return self.sae_out[:n], self.topk_idx[:n], self.topk_val[:n]

    # ---- feature-parallel step (this engine = one rank's feature shard; sae/feature_parallel.py drives it) -----------
    def tp_partial(self, idx: torch.Tensor, val: torch.Tensor, renorm_decoder: bool = True) -> torch.Tensor:
        """[N, d_in] partial reconstruction of this shard's kept pairs (value 0 = not kept); a view, overwritten by the
        next call."""
        self._no_pending_sparse("tp_partial")
        n = idx.shape[0]
        assert idx.dtype == torch.int32 and val.dtype == torch.float32 and idx.is_contiguous() and val.is_contiguous()
        st = self._state()
        inv_valid = renorm_decoder and self._inv_norm_key is not None and self._inv_norm_key == self._w_dec_key()
        N.check(self.lib.pv_sae_tp_partial(self._plan, C.byref(st), idx.data_ptr(), val.data_ptr(), n,
                                           (2 if renorm_decoder else 0) | (4 if inv_valid else 0), self.sae_out.data_ptr(),
                                           self._stream()), "pv_sae_tp_partial")
        return self.sae_out[:n]

    def tp_finish(self, x: torch.Tensor, pre_sum: torch.Tensor, idx: torch.Tensor, val: torch.Tensor, n_global: Optional[int] = None,
                  enc_term_only: bool = False, update_stats: bool = False) -> None:
        """Everything behind the summed reconstruction ``pre_sum`` (no b_dec): loss, gradients of this shard's rows (``g``),
        ``g['b_dec']`` = colsum(dY) - this shard's encoder term (without colsum(dY) when enc_term_only), scalars[0..1] =
        loss, scalars[2] = this shard's kept pairs per token.  Must follow ``encode_topk(x)`` on the same x."""
        x = self._check_x(x)
        n = x.shape[0]
        pre_sum = pre_sum.contiguous()
        assert pre_sum.dtype == torch.float32 and tuple(pre_sum.shape) == (n, self.d_in)
        st = self._state()
        out = N.SaeOut(sae_out=None, topk_idx=idx.data_ptr(), topk_val=val.data_ptr(), scalars=self.scalars.data_ptr(),
                       fire_count=self.fire_count.data_ptr())
        N.check(self.lib.pv_sae_tp_finish(self._plan, C.byref(st), x.data_ptr(), pre_sum.data_ptr(), idx.data_ptr(), val.data_ptr(),
                                          n, int(n_global if n_global is not None else n),
                                          int(bool(update_stats)) | (16 if enc_term_only else 0), C.byref(out),
                                          self.workspace.data_ptr(), self.workspace.numel(), self._stream()), "pv_sae_tp_finish")
        self._grad_fresh = False            # (grad_sqnorm(from_step=True) is about pv_sae_step; use grad_sqnorm_rows here)
        self._sq_fused = False
        self._grad_sparse = False

    # the feature-parallel step's glue (sae/feature_parallel.py): exchange buffers the kernels write in place
    def tp_bind(self, pack: torch.Tensor, bucket: torch.Tensor, lo: int, d_sae_total: int) -> None:
        """Point this shard engine's outputs into the step's two exchange buffers: the candidates of ``encode_topk`` land
        in ``pack`` [2, max_tokens, k] int32 (values as float bits | local indices: ONE all-gather), ``tp_finish`` writes its
        ``gb_dec`` term into ``bucket[:d_in]`` and the shard's firing counts into ``bucket[d_in + 4 + lo:...]`` (ONE small
        all-reduce, see ``tp_bucket_pack``)."""
        assert pack.dtype == torch.int32 and tuple(pack.shape) == (2, self.max_tokens, self.k) and pack.is_contiguous()
        assert bucket.dtype == torch.float32 and bucket.numel() == self.d_in + 4 + d_sae_total and bucket.is_contiguous()
        self.topk_val = pack[0].view(torch.float32)
        self.topk_idx = pack[1]
        self._g["b_dec"] = self.g["b_dec"] = bucket[:self.d_in]
        self.fire_count = bucket[self.d_in + 4 + lo:self.d_in + 4 + lo + self.d_sae]
        self._tp = (int(lo), int(d_sae_total))
        self._tp_val_kept = torch.zeros(self.max_tokens, self.k, dtype=torch.float32, device=self.device)

    def tp_merge(self, gathered: torch.Tensor, world: int, rank: int, n: int) -> torch.Tensor:
        """gathered [world, 2, n, k] int32 (the all-gather of the ranks' ``pack[:, :n]``) -> this rank's candidate values
        where they are among the k best of all ranks' candidates of their token, 0 elsewhere (pv_sae_tp_merge)."""
        assert gathered.dtype == torch.int32 and gathered.is_contiguous() and tuple(gathered.shape) == (world, 2, n, self.k)
        out = self._tp_val_kept[:n]
        N.check(self.lib.pv_sae_tp_merge(gathered.data_ptr(), int(world), int(rank), int(n), self.k, self.d_sae, out.data_ptr(),
                                         self._stream()), "pv_sae_tp_merge")
        return out

    def tp_bucket_pack(self, bucket: torch.Tensor) -> None:
        lo, total = self._tp
        N.check(self.lib.pv_sae_tp_bucket_pack(self._plan, self.workspace.data_ptr(), self.scalars.data_ptr(), bucket.data_ptr(),
                                               lo, total, self._stream()), "pv_sae_tp_bucket_pack")

    def tp_bucket_unpack(self, bucket: torch.Tensor) -> None:
        N.check(self.lib.pv_sae_tp_bucket_unpack(self._plan, bucket.data_ptr(), self.scalars.data_ptr(), self._stream()),
                "pv_sae_tp_bucket_unpack")

    # convenience: one full reference train_step (train_sae.py:278-411) on a single GPU
    def train_step(self, x: torch.Tensor, lr: float, max_grad_norm: Optional[float] = 1.0) -> None:
        self.step(x, renorm_decoder=True, sparse_grads=True)
        self.grad_sqnorm(from_step=True)
        self.apply(lr, max_grad_norm)

    def grad_W_enc(self) -> torch.Tensor:
        """Gradient of W_enc in the parameter's own [d_in, d_sae] layout (a transposed view)."""
        return self._g["W_encT"].t()

    def fallback_rows(self) -> int:
        """Tokens of the last encode the filter could not decide (recomputed exactly); a device read-back, for tests."""
        if not self.filtered_encoder:
            return 0
        off = self.lib.pv_debug_sae_ws_offset(self._plan, b"fb_count")
        return int(self.workspace[off:off + 4].view(torch.int32).item())
