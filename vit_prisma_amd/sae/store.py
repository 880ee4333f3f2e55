"""VisionActivationsStore: streams images through the ViT and serves shuffled token batches.

Same constructor, attributes and buffer algorithm as the reference's
/root/reference/src/vit_prisma/sae/training/activations_store.py:176-503 (half-buffer shuffle-mix,
inner DataLoader of ``[train_batch_size, n_layers, d_in]`` batches).  The producer call is
``model.run_with_cache(batch, names_filter=[hook_point...], stop_at_layer=L+1)`` under ``no_grad`` --
on an MI355X that dispatches to the native HIP plan, which runs only blocks 0..L and writes the single
requested activation straight into the tap slab.

Data-parallel harvesting: with ``torch.distributed`` initialised every rank
draws a disjoint shard of each epoch's images (DistributedSampler) and fills its own local buffer; no
collective is involved in harvesting.
"""
from __future__ import annotations

import os
from typing import Any, Iterator, List, Optional

import torch
from torch.utils.data import DataLoader


def collate_fn(data):
    return torch.stack([d[0] for d in data], dim=0)


def collate_fn_eval(data):
    return torch.stack([d[0] for d in data], dim=0), torch.tensor([d[1] for d in data])


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class VisionActivationsStore:
    def __init__(self, cfg, model, dataset, create_dataloader: bool = True, eval_dataset=None, num_workers: int = 0):
        self.cfg = cfg
        self.model = model.to(cfg.device)
        self.dataset = dataset
        self.n_tokens_harvested = 0                          # rows written into buffers so far (bench: harvested / trained)
        self.n_buffer_copies = 0                             # `buf[...] = acts` passes taken (0 while the producing kernel writes the rows itself)
        # The next refill's ViT forwards do not depend on the SAE being trained: on a GPU they are issued on a side stream as
        # soon as the current half buffer is being served and overlap the train steps (whose many small kernels leave most of
        # the chip idle); the refill then only waits for an event.  Same images, same order, same random permutations drawn
        # at the same points -- the batches served are identical to the synchronous store's.  ``overlap_harvest = False``
        # turns it off.
        self.overlap_harvest = torch.device(cfg.device).type == "cuda"
        self.direct_tap = True                               # harvest kernels store straight into the buffer (see _harvest_raw)
        self._prefetched = None                              # (n_batches, buf, buf_out, event) of the refill in flight
        self._side_stream = None
        rank, world = _dist_info()
        sampler = None
        if world > 1:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
        self.image_dataloader = DataLoader(dataset, shuffle=sampler is None, sampler=sampler, num_workers=num_workers,
                                           batch_size=cfg.store_batch_size, collate_fn=collate_fn, drop_last=True)
        if eval_dataset is not None:
            self.image_dataloader_eval = DataLoader(eval_dataset, shuffle=True, num_workers=num_workers,
                                                    batch_size=cfg.store_batch_size, collate_fn=collate_fn_eval,
                                                    drop_last=True)
            self.image_dataloader_eval_iter = self._eval_batch_stream(self.image_dataloader_eval, cfg.device)
        self.image_dataloader_iter = self._batch_stream(self.image_dataloader, cfg.device)
        if create_dataloader:
            if cfg.is_transcoder:
                self.storage_buffer, self.storage_buffer_out = self.get_buffer(cfg.n_batches_in_buffer)
            else:
                self.storage_buffer = self.get_buffer(cfg.n_batches_in_buffer)
            self.dataloader = self.get_data_loader()

    # ---- image streams ----------------------------------------------------------------------------
    def _batch_stream(self, dataloader: DataLoader, device) -> Iterator[torch.Tensor]:
        epoch = 0
        while True:
            if hasattr(dataloader.sampler, "set_epoch"):
                dataloader.sampler.set_epoch(epoch)
            for batch in dataloader:
                batch.requires_grad_(False)
                yield batch.to(device, non_blocking=True)
            epoch += 1

    def _eval_batch_stream(self, dataloader: DataLoader, device):
        while True:
            for images, labels in dataloader:
                yield images.to(device), labels.to(device)

    # ---- harvesting -------------------------------------------------------------------------------
    def _layers(self) -> List[int]:
        hl = self.cfg.hook_point_layer
        return list(hl) if isinstance(hl, list) else [hl]

    @torch.no_grad()
    def get_activations(self, batch_tokens: torch.Tensor):
        """[B, ctx, n_layers, d_in] activations of the configured hook point(s); for a transcoder the pair
        (input hook point(s), output hook point(s)) harvested by ONE forward (activations_store.py:251-296)."""
        cfg = self.cfg
        layers = self._layers()
        if isinstance(cfg.hook_point_layer, list):
            names = [cfg.hook_point.format(layer=layer) for layer in layers]
        else:
            names = [cfg.hook_point]
        out_names, stop = [], max(layers) + 1
        if cfg.is_transcoder:
            ol = cfg.out_hook_point_layer
            out_layers = list(ol) if isinstance(ol, list) else [ol]
            out_names = [cfg.out_hook_point] if not isinstance(ol, list) else [
                f"blocks.{layer}.{cfg.layer_out_subtype}" for layer in out_layers]
            stop = max(max(layers), max(out_layers)) + 1
        _, cache = self.model.run_with_cache(batch_tokens, names_filter=names + out_names, stop_at_layer=stop)

        def pick(which):
            acts = []
            for name in which:
                a = cache[name]
                if cfg.hook_point_head_index is not None:
                    a = a[:, :, cfg.hook_point_head_index]
                if cfg.cls_token_only:
                    a = a[:, 0:1]
                acts.append(a)
            if len(acts) == 1:
                return acts[0].unsqueeze(2)                   # (a view: torch.stack would copy -- and hide from _harvest_raw that the kernel
            return torch.stack(acts, dim=2)                   #  already wrote these rows into the buffer slice)

        if cfg.is_transcoder:
            return pick(names), pick(out_names)
        return pick(names)

    # images per ViT forward while a buffer is harvested: the reference runs one forward per store batch (32 images by default,
    # config.py:352) -- ~100 launch-bound kernels each on this hardware.  Consecutive store batches are independent images, and
    # the HIP forward's bits do not depend on the batch an image sits in (tests: digests at bs 1 / 3 / 77 / 300 / 512), so
    # up to HARVEST_IMAGES / store_batch_size DataLoader batches go through ONE run_with_cache: same images, same order, same
    # buffer rows, same bits.  (512: a half-buffer refill of the reference's default store shape -- 10 batches of 32 -- is one forward.)
    HARVEST_IMAGES = 512

    def _direct_tap_name(self) -> Optional[str]:
        """The single hook point whose [images, ctx, d_in] activation IS a buffer slice, or None (several layers, a head index,
        CLS only, patches only, a transcoder: those assemble the rows in ``get_activations``)."""
        cfg = self.cfg
        if (cfg.is_transcoder or isinstance(cfg.hook_point_layer, list) or cfg.hook_point_head_index is not None
                or cfg.cls_token_only or cfg.use_patches_only):
            return None
        return cfg.hook_point

    def _harvest_raw(self, n_batches_in_buffer: int):
        """The ViT part of ``get_buffer``: (buf [bs * n_batches, ctx, n_layers, d_in], buf_out or None), rows in harvest order.
        Where the activation of the one configured hook point is exactly a slice of the buffer, the producing kernel stores it
        THERE (``tap_dst``: no arena copy, no ``buf[...] = acts`` pass); the buffer is then held in the ViT's activation dtype
        (bf16 under a bf16 ViT: the values the reference's assignment would widen, bit for bit) and ``get_buffer`` widens the
        shuffled rows to cfg.dtype."""
        cfg = self.cfg
        bs = cfg.store_batch_size
        total = bs * n_batches_in_buffer
        n_layers = len(self._layers())
        ctx = cfg.context_size
        direct = self._direct_tap_name()
        model_dtype = getattr(getattr(self.model, "cfg", None), "dtype", None)
        on_hip = getattr(self.model, "native_mode", "off") != "off" and torch.device(cfg.device).type == "cuda"
        native = direct is not None and on_hip and self.direct_tap and model_dtype in (torch.float32, torch.bfloat16)
        buf = torch.zeros((total, ctx, n_layers, cfg.d_in), dtype=model_dtype if native else cfg.dtype, device=cfg.device)
        buf_out = None
        if cfg.is_transcoder:
            ol = cfg.out_hook_point_layer
            buf_out = torch.zeros((total, ctx, len(ol) if isinstance(ol, list) else 1, cfg.d_out), dtype=cfg.dtype, device=cfg.device)
        group = max(1, self.HARVEST_IMAGES // bs) if on_hip else 1      # (the PyTorch path's matmuls are not batch-size invariant bit for bit)
        # the ViT is constant while one buffer is harvested: skip the per-call weight-version compare inside this loop
        # only (an edit of the model between buffers is picked up by the next one)
        freeze = getattr(self.model, "freeze_native_weights", None)
        was_frozen = bool(getattr(self.model, "_native_frozen", False))
        if freeze is not None:
            freeze(True)
        try:
            start = 0
            while start < total:
                g = min(group, (total - start) // bs)
                images = [next(self.image_dataloader_iter) for _ in range(g)]
                images = images[0] if g == 1 else torch.cat(images, dim=0)
                rows = g * bs
                dst = buf[start:start + rows]
                if native:
                    self.model._tap_dst = {direct: dst.view(rows, ctx, cfg.d_in)}
                try:
                    acts = self.get_activations(images)
                finally:
                    if native:
                        self.model._tap_dst = None
                acts_out = None
                if cfg.is_transcoder:
                    acts, acts_out = acts
                if cfg.use_patches_only:
                    acts = acts[:, 1:, :, :]
                    acts_out = acts_out[:, 1:, :, :] if acts_out is not None else None
                if acts_out is not None:
                    buf_out[start:start + rows, ...] = acts_out
                # the reference's assignment (activations_store.py:345): a [bs, 1, L, d] CLS-only harvest broadcasts over
                # context_size, any other token-count mismatch raises.  Skipped when the kernel already stored the rows here.
                if not (native and acts.data_ptr() == dst.data_ptr() and acts.dtype == dst.dtype and acts.shape == dst.shape):
                    dst[...] = acts
                    self.n_buffer_copies += 1
                self.n_tokens_harvested += rows * ctx
                start += rows
        finally:
            if freeze is not None:
                freeze(was_frozen)
        return buf, buf_out

    def _start_prefetch(self, n_batches: int) -> None:
        """Issue the next refill's harvest on the side stream (returns at once; ``get_buffer`` picks it up)."""
        dev = torch.device(self.cfg.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=dev)
        side, cur = self._side_stream, torch.cuda.current_stream(dev)
        side.wait_stream(cur)                                 # (whatever the caller did to the model / images so far)
        with torch.cuda.stream(side):
            buf, buf_out = self._harvest_raw(n_batches)
            ev = torch.cuda.Event()
            ev.record(side)
        self._prefetched = (n_batches, buf, buf_out, ev)

    def get_buffer(self, n_batches_in_buffer: int) -> torch.Tensor:
        """[bs * n_batches * ctx, n_layers, d_in], rows shuffled (activations_store.py:298-362)."""
        cfg = self.cfg
        n_layers = len(self._layers())
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] == n_batches_in_buffer:
            _, buf, buf_out, ev = pre
            cur = torch.cuda.current_stream(buf.device)
            cur.wait_event(ev)
            buf.record_stream(cur)                            # (allocated on the side stream, consumed and freed on this one)
            if buf_out is not None:
                buf_out.record_stream(cur)
        else:
            if pre is not None:                               # a prefetch of another size: its images are spent, keep the order
                torch.cuda.current_stream(pre[1].device).wait_event(pre[3])
            buf, buf_out = self._harvest_raw(n_batches_in_buffer)
        buf = buf.reshape(-1, n_layers, cfg.d_in)
        perm = torch.randperm(buf.shape[0], device=buf.device)
        if buf_out is not None:
            return buf[perm].to(cfg.dtype), buf_out.reshape(-1, buf_out.shape[2], cfg.d_out)[perm]
        return buf[perm].to(cfg.dtype)                            # (a no-op unless the rows were tapped in the ViT's narrower dtype)

    def get_data_loader(self) -> Iterator[Any]:
        """Mix a fresh half buffer into the stored one, keep half, serve the other half
        (activations_store.py:445-492).  Transcoder: input and target rows are shuffled together and served side by
        side along dim 1 (``batch[:, 0]`` input, ``batch[:, 1]`` target; requires d_out == d_in like the reference's cat)."""
        cfg = self.cfg
        if cfg.is_transcoder:
            new, new_out = self.get_buffer(cfg.n_batches_in_buffer // 2)
            mix = torch.cat([new, self.storage_buffer], dim=0)
            mix_out = torch.cat([new_out, self.storage_buffer_out], dim=0)
            perm = torch.randperm(mix.shape[0], device=mix.device)
            mix, mix_out = mix[perm], mix_out[perm]
            half = mix.shape[0] // 2
            self.storage_buffer, self.storage_buffer_out = mix[:half], mix_out[:half]
            serve = torch.cat([mix[half:], mix_out[half:]], dim=1)
        else:
            mix = torch.cat([self.get_buffer(cfg.n_batches_in_buffer // 2), self.storage_buffer], dim=0)
            mix = mix[torch.randperm(mix.shape[0], device=mix.device)]
            half = mix.shape[0] // 2
            self.storage_buffer = mix[:half]
            serve = mix[half:]
        _, world = _dist_info()
        local_bs = max(cfg.train_batch_size // world, 1)
        if self.overlap_harvest and cfg.n_batches_in_buffer // 2 > 0:
            self._start_prefetch(cfg.n_batches_in_buffer // 2)    # the refill this iterator will end in, behind the train steps
        return iter(_TensorBatches(serve, local_bs))

    def next_batch(self) -> torch.Tensor:
        try:
            return next(self.dataloader)
        except StopIteration:
            self.dataloader = self.get_data_loader()
            return next(self.dataloader)


    # ---- on-disk activation cache -----------------------------------------------
    @torch.no_grad()
    def generate_cached_activations_from_dataset(self, tokens_per_file: int = 1_000_000, shuffle_data: bool = False) -> int:
        """Harvest the whole dataset once and write ``{idx}.pt`` shards of ``[tokens, n_layers, d_in]`` **fp16**
        tensors (``tokens_per_file`` rows each, the last one shorter) under ``cfg.cached_activations_path`` -- the
        format ``CacheVisionActivationStore`` (here and in the reference) reads back
        (activations_store.py:505-575; ``.half()`` :546).  The producer is the same native
        ``run_with_cache(names_filter, stop_at_layer)`` call as live harvesting; a shard is assembled in HBM and
        leaves the GPU in one copy.  Returns the number of files written."""
        cfg = self.cfg
        save_dir = cfg.cached_activations_path
        os.makedirs(save_dir, exist_ok=True)
        loader = DataLoader(self.dataset, batch_size=cfg.store_batch_size, shuffle=shuffle_data, num_workers=0,
                            drop_last=False)
        n_layers = len(self._layers())
        pending: List[torch.Tensor] = []
        n_pending = 0
        file_idx = 0

        def flush(rows: torch.Tensor) -> None:
            nonlocal file_idx
            torch.save(rows.cpu(), os.path.join(save_dir, f"{file_idx}.pt"))
            file_idx += 1

        for batch in loader:
            images = batch[0] if isinstance(batch, (list, tuple)) else batch
            acts = self.get_activations(images.to(cfg.device)).half()
            if cfg.use_patches_only:
                acts = acts[:, 1:, :, :]
            flat = acts.reshape(-1, n_layers, cfg.d_in)
            pending.append(flat)
            n_pending += flat.shape[0]
            while n_pending >= tokens_per_file:
                combined = torch.cat(pending, dim=0)
                flush(combined[:tokens_per_file])
                combined = combined[tokens_per_file:]
                n_pending = combined.shape[0]
                pending = [combined] if n_pending > 0 else []
        if n_pending > 0:
            flush(torch.cat(pending, dim=0))
        return file_idx


class CacheVisionActivationStore:
    """Serves training batches from the ``{idx}.pt`` shards instead of a live model
    (activations_store.py:21-152): same constructor (``cfg`` only, ``cfg.use_cached_activations`` must be set), same
    ``storage_buffer`` / ``get_buffer`` / ``get_data_loader`` / ``next_batch`` and the same half-buffer shuffle-mix.
    Like the reference, every refill starts reading at shard 0 (its ``next_cache_idx`` is a local of
    ``_load_cached_activations``): the cache is meant to be at least one buffer long.
    Data parallel (no reference counterpart): rank r of ``world`` reads the shards r, r + world, ... and serves
    ``train_batch_size // world`` tokens per step, like the live store's DistributedSampler split."""

    def __init__(self, cfg, rank: int = 0, world: int = 1):
        self.cfg = cfg
        self.rank, self.world = int(rank), max(int(world), 1)
        if not cfg.use_cached_activations:
            raise ValueError("CacheVisionActivationStore cannot be initialized with cfg.use_cached_activations = False ")
        self._files = {}
        self.next_idx_within_buffer = 0
        half = cfg.n_batches_in_buffer // 2
        self.storage_buffer = self.get_buffer(half)
        self.dataloader = self.get_data_loader()

    def load_file_cached(self, file: str) -> torch.Tensor:
        if file not in self._files:
            if len(self._files) >= 2:                       # the reference keeps an lru_cache(maxsize=2)
                self._files.pop(next(iter(self._files)))
            self._files[file] = torch.load(file)
        return self._files[file]

    def _load_cached_activations(self, total_size: int, context_size: int, num_layers: int, d_in: int) -> torch.Tensor:
        cfg = self.cfg
        buffer_size = total_size * context_size
        buf = torch.zeros((buffer_size, num_layers, d_in), dtype=cfg.dtype, device=cfg.device)
        filled = 0
        idx = self.rank
        while filled < buffer_size:
            path = f"{cfg.cached_activations_path}/{idx}.pt"
            if not os.path.exists(path):
                return buf[:filled]
            acts = self.load_file_cached(path)
            partial = filled + acts.shape[0] > buffer_size
            if partial:
                acts = acts[: buffer_size - filled]
            buf[filled:filled + acts.shape[0]] = acts.to(cfg.device)
            filled += acts.shape[0]
            if partial:
                self.next_idx_within_buffer = acts.shape[0]
            else:
                idx += self.world
                self.next_idx_within_buffer = 0
        return buf

    def get_buffer(self, n_batches_in_buffer: int) -> torch.Tensor:
        cfg = self.cfg
        n_layers = len(cfg.hook_point_layer) if isinstance(cfg.hook_point_layer, list) else 1
        return self._load_cached_activations(cfg.store_batch_size * n_batches_in_buffer, cfg.context_size, n_layers, cfg.d_in)

    def get_data_loader(self) -> Iterator[Any]:
        cfg = self.cfg
        mix = torch.cat([self.get_buffer(cfg.n_batches_in_buffer // 2), self.storage_buffer], dim=0)
        mix = mix[torch.randperm(mix.shape[0], device=mix.device)]
        half = mix.shape[0] // 2
        self.storage_buffer = mix[:half]
        return iter(_TensorBatches(mix[half:], max(cfg.train_batch_size // self.world, 1)))

    def next_batch(self) -> torch.Tensor:
        try:
            return next(self.dataloader)
        except StopIteration:
            self.dataloader = self.get_data_loader()
            return next(self.dataloader)


class _TensorBatches:
    """Shuffled fixed-size batches of a device-resident tensor (what ``DataLoader(tensor,
    batch_size, shuffle=True)`` yields in the reference, without the per-row Python collate; the last
    partial batch is served too, like DataLoader's default ``drop_last=False``)."""

    def __init__(self, data: torch.Tensor, batch_size: int):
        self.data = data
        self.batch_size = batch_size

    def __iter__(self):
        perm = torch.randperm(self.data.shape[0], device=self.data.device)
        for i in range(0, self.data.shape[0], self.batch_size):
            yield self.data[perm[i:i + self.batch_size]]
