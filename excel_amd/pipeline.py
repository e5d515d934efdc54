"""Batched, fully device-resident training-free pipeline over the same HIP kernels the per-image API uses.

One `run_batch` = tools/infer_lam.py:74-114 for B images at once:
    ViT surgery forward (+ fused affinity mean) -> patch-text CAM -> box-masked attention random walk
    -> min-max + cv2-style up-sampling + background -> PAR x20 -> arg-max -> confusion accumulate.
No host round trip happens inside a step (the reference makes ~2k+5 per image).
"""
import torch

from . import ops


class TrainingFreePipeline:
    def __init__(self, model, num_classes=21, dilations=ops.PAR_DILATIONS, num_iter=20, caa_thre=0.79, smax=6):
        """`smax` = the largest number of present classes of any image that will be fed (known from the host-side
        image-level labels; VOC train_aug: 6)."""
        self.model = model
        self.num_classes = num_classes
        self.dilations = tuple(dilations)
        self.num_iter = num_iter
        self.caa_thre = caa_thre
        self.smax = smax
        self.hist = None
        self._bufs = {}

    def reset(self):
        self.drain()
        self.hist = None

    def _buf(self, name, numel, dtype=torch.float32, device=None):
        """Step-persistent scratch (cams, PAR output, PAR workspace): one grow-only flat buffer per (name, launch stream), so a step
        allocates nothing and launches no fill kernels.  Only buffers nobody outside the step keeps are taken from here.
        The buffers are NOT initialised and the producers skip what nobody reads: pad columns (W_b <= x < Wp_b) and the planes of
        unused channels (c >= nchan[b]) of the step's cams / PAR output hold garbage.  The step's own consumers clamp to W_b - 1 and
        nchan[b]; a new consumer (saving cams, a CRF stage) must do the same or ask for `return_intermediates=True`, which switches to
        fresh zero-filled tensors."""
        key = (name, torch.cuda.current_stream().cuda_stream)
        b = self._bufs.get(key)
        if b is None or b.numel() < numel or b.dtype != dtype:
            self._bufs.pop(key, None)
            b = None
            b = self._bufs[key] = torch.empty(int(numel), dtype=dtype, device=device)
        return b[:int(numel)]

    @torch.no_grad()
    def run_batch(self, inputs, cls_labels, gts=None, label_hw=None, return_intermediates=False):
        """inputs [B,3,S,S] f32 (normalised, already at the network size), cls_labels [B,F] f32 one-hot,
        gts [B,H,W] uint8 (255 = ignore) or None.  Returns labels [B,H,W] uint8 (device)."""
        B, _, S, _ = inputs.shape
        g = S // 16
        H, W = (gts.shape[-2:] if gts is not None else (label_hw or (S, S)))
        _, _, attr, attn_w, _ = self.model(inputs)                                                  # infer_lam.py:79
        idx, ncls, nchan = ops.cls_compact(cls_labels, self.smax, want_nchan=True)                  # affutils.py:203
        refined = ops.refine_cams_with_aff_batched(attr, attn_w.w_aff, idx, ncls, g, self.caa_thre)  # infer_lam.py:93
        C = self.smax + 1
        if return_intermediates:            # the caller keeps these: fresh tensors, unused channels zeroed
            cams = ops.cam_upsample_bkg(refined, ncls, g, H, W)                                     # affutils.py:164-166
            par_out = ops.par_forward(inputs, cams, self.dilations, self.num_iter, nchan=nchan)    # affutils.py:84
        else:
            dev = inputs.device
            cams = ops.cam_upsample_bkg(refined, ncls, g, H, W, out=self._buf("cams", B * C * H * W, device=dev).view(B, C, H, W),
                                        zero_unused=False)
            ws = self._buf("par_ws", ops.lib().excel_par_workspace_bytes(B, C, H, W, len(self.dilations)), torch.uint8, dev)
            par_out = ops.par_forward(inputs, cams, self.dilations, self.num_iter, nchan=nchan, ws=ws,
                                      out=self._buf("par_out", B * C * H * W, device=dev).view(B, C, H, W))
        labels = ops.argmax_label(par_out, nchan, idx)                                              # affutils.py:86-87
        if gts is not None:
            self.hist = ops.confusion_accumulate(gts, labels, self.num_classes, self.hist)          # evaluate.py:9-20
        if return_intermediates:
            return labels, dict(attr=attr, w_aff=attn_w.w_aff, refined=refined, cams=cams, par_out=par_out,
                                cls_idx=idx, ncls=ncls)
        return labels

    # ------------------------------------------------------------------ ragged batches: every image at its OWN label size
    # The reference resizes the input to S x S but refines and scores at the image's original size (tools/infer_lam.py:74,94:
    # labels.shape[-2:]), batch 1.  Here the size-uniform half (ViT, CAM, random walk) runs as one batch as before, and the
    # size-dependent half (input resize, up-sampling, PAR, arg-max) runs over packed, pitched planes through a tile map
    # (ops.RaggedPlan; include/excel_hip.h "ragged batches"): one launch per stage whatever the mix of sizes.
    @torch.no_grad()
    def run_batch_ragged(self, hwc_packed, plan, cls_labels, gts_packed=None, S=448, return_intermediates=False):
        """hwc_packed: the decoded uint8 [H_b,W_b,3] images back to back (device); plan = ops.RaggedPlan of their sizes;
        cls_labels [B,F] f32 one-hot; gts_packed: the uint8 [H_b,W_b] ground-truth maps back to back (255 = ignore) or None.
        Returns the labels as one flat uint8 tensor (image b = plan.label(labels, b)).
        Without `return_intermediates` the cams / PAR output live in uninitialised step buffers (see _buf: pad columns and unused
        channels are undefined); with it they are fresh tensors whose unused parts are zero."""
        dev = hwc_packed.device
        B = plan.B
        g = S // 16
        inputs = ops.normalize_resize_u8_ragged(hwc_packed, plan, S, out=self._buf("inputs", B * 3 * S * S, device=dev).view(B, 3, S, S))  # voc.py:115, infer_lam.py:74
        _, _, attr, attn_w, _ = self.model(inputs)                                                  # infer_lam.py:79
        idx, ncls, nchan = ops.cls_compact(cls_labels, self.smax, want_nchan=True)                  # affutils.py:203
        refined = ops.refine_cams_with_aff_batched(attr, attn_w.w_aff, idx, ncls, g, self.caa_thre)  # infer_lam.py:93
        C = self.smax + 1
        keep = return_intermediates
        cams = ops.cam_upsample_bkg_ragged(refined, ncls, g, plan, zero_unused=keep,
                                           out=None if keep else self._buf("cams", C * plan.total_pix, device=dev))   # affutils.py:164-166
        ws = self._buf("par_ws", ops.lib().excel_par_ragged_workspace_bytes(plan.total_pix, C), torch.uint8, dev)
        par_out = ops.par_forward_ragged(inputs, cams, plan, C, self.dilations, self.num_iter, nchan=nchan, ws=ws,
                                         out=None if keep else self._buf("par_out", C * plan.total_pix, device=dev))   # affutils.py:84
        labels = ops.argmax_label_ragged(par_out, plan, C, nchan, idx)                              # affutils.py:86-87
        if gts_packed is not None:
            self.hist = ops.confusion_accumulate(gts_packed, labels, self.num_classes, self.hist)   # evaluate.py:9-20
        if return_intermediates:
            return labels, dict(inputs=inputs.clone(), attr=attr, w_aff=attn_w.w_aff, refined=refined, cams=cams, par_out=par_out,
                                cls_idx=idx, ncls=ncls)
        return labels

    # ------------------------------------------------------------------ concurrent sub-batches
    # Every kernel of the path leaves part of the chip idle in its last round of workgroups (e.g. the N=768 GEMMs of a
    # 32-image batch are 594 tiles for 256 CUs: 2.3 rounds).  run_batch_split runs the batch as `nsplit` independent
    # sub-batches, each on its own stream with its own workspaces: the tail of one sub-batch's kernel is filled by the
    # other's.  Same kernels, per-image results bit-identical to run_batch (images are independent).  Measured at B=32,
    # nsplit=2: +6 % before the GEMM tile selection became round-aware (gemm_bf16x3.hip), +1 % after; nsplit=4: no gain.
    @torch.no_grad()
    def run_batch_split(self, inputs, cls_labels, gts=None, label_hw=None, nsplit=2):
        """Same contract as run_batch; the returned labels and self.hist are complete only after drain()."""
        B, _, S, _ = inputs.shape
        nsplit = max(1, min(nsplit, B))
        if nsplit == 1:
            return self.run_batch(inputs, cls_labels, gts, label_hw)
        if len(getattr(self, "_sub", ())) != nsplit:
            self._sub = [dict(stream=torch.cuda.Stream(), hist=None) for _ in range(nsplit)]
        cur = torch.cuda.current_stream()
        H, W = (gts.shape[-2:] if gts is not None else (label_hw or (S, S)))
        labels = torch.empty((B, H, W), dtype=torch.uint8, device=inputs.device)
        step = -(-B // nsplit)
        for i, sub in enumerate(self._sub):
            lo, hi = i * step, min(B, (i + 1) * step)
            if lo >= hi:
                continue
            st = sub["stream"]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for t in (inputs, cls_labels, labels) + ((gts,) if gts is not None else ()):
                    t.record_stream(st)
                main_hist, self.hist = self.hist, sub["hist"]
                try:
                    part = self.run_batch(inputs[lo:hi], cls_labels[lo:hi], None if gts is None else gts[lo:hi], (H, W))
                    sub["hist"] = self.hist
                finally:
                    self.hist = main_hist
                labels[lo:hi].copy_(part)
        return labels

    # ------------------------------------------------------------------ two-stream software pipeline
    # Stage A (ViT + CAM + random walk + up-sampling: matrix-core bound) and stage B (PAR + argmax + confusion:
    # HBM bound) use different hardware resources.  run_batch_overlapped enqueues stage A of batch i on stream A and
    # its stage B on stream B behind an event, so stage B of batch i overlaps stage A of batch i+1.  Results are
    # identical to run_batch (same kernels, same order per batch); only the issue order across batches changes.
    def _streams(self):
        if not hasattr(self, "_sa"):
            self._sa = torch.cuda.Stream()
            self._sb = torch.cuda.Stream()
            self._sb_done = None
        return self._sa, self._sb

    @torch.no_grad()
    def run_batch_overlapped(self, inputs, cls_labels, gts=None, label_hw=None):
        sa, sb = self._streams()
        cur = torch.cuda.current_stream()
        B, _, S, _ = inputs.shape
        g = S // 16
        H, W = (gts.shape[-2:] if gts is not None else (label_hw or (S, S)))
        sa.wait_stream(cur)
        with torch.cuda.stream(sa):
            # the caller may free / reuse these right after the call: the caching allocator must know stream A reads them
            inputs.record_stream(sa)
            cls_labels.record_stream(sa)
            _, _, attr, attn_w, _ = self.model(inputs)
            idx, ncls, nchan = ops.cls_compact(cls_labels, self.smax, want_nchan=True)
            refined = ops.refine_cams_with_aff_batched(attr, attn_w.w_aff, idx, ncls, g, self.caa_thre)
            cams = ops.cam_upsample_bkg(refined, ncls, g, H, W)
            ready = torch.cuda.Event()
            ready.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(ready)
            for t in (cams, nchan, idx, inputs) + ((gts,) if gts is not None else ()):
                t.record_stream(sb)
            par_out = ops.par_forward(inputs, cams, self.dilations, self.num_iter, nchan=nchan)
            labels = ops.argmax_label(par_out, nchan, idx)
            if gts is not None:
                self.hist = ops.confusion_accumulate(gts, labels, self.num_classes, self.hist)
        return labels

    def drain(self):
        """Make the caller's stream wait for the pipeline's side streams and fold the sub-batch histograms into
        self.hist (call before reading hist / labels of run_batch_split / run_batch_overlapped)."""
        cur = torch.cuda.current_stream()
        if hasattr(self, "_sa"):
            cur.wait_stream(self._sa)
            cur.wait_stream(self._sb)
        for sub in getattr(self, "_sub", ()):
            cur.wait_stream(sub["stream"])
            if sub["hist"] is not None:
                self.hist = sub["hist"] if self.hist is None else self.hist + sub["hist"]
                sub["hist"] = None
