"""DINO-style VisionTransformer whose forward/backward run on libvitmi HIP kernels.

Drop-in for the module the reference obtains from
`torch.hub.load('facebookresearch/dino:main', arch)`
(/root/reference/models/vision_all.py:156): same constructor arguments,
attribute surface (`.patch_embed.proj` Conv2d, `.norm`, `.head`, `.blocks[i]`),
parameter names/shapes (state-dict compatible) and call contract
`model(x[B,C,H,W] fp32) -> [B, K or D] fp32` (utils_network.py:418).

The nn.Linear / nn.LayerNorm / nn.Conv2d sub-modules are parameter holders
only; `forward` hands the whole network to `VitEngine`, which sequences the
hand-written kernels (GEMM+epilogues, LayerNorm, fused attention, ...) through
the C ABI and implements the backward pass explicitly (the reference relies on
autograd, utils_network.py:441).  There is no PyTorch fallback.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import (EPI_BIAS_GELU, EPI_DGELU, EPI_PATCH_POS, EPI_RESIDUAL, EPI_STORE, GEMM_AUTO,
                   LAUNCH_ROWS_PADDED, VitmiError)
from .packing import ParamPack
from .posembed import tables_for
from .data import PatchRows

# compute_dtype: "bf16" (bf16 operands, fp32 accumulate: the benchmarked mode), "fp32" (fp32 MFMA / VALU: the exact parity
# mode) or "bf16x3" (round 5: fp32 activations and weights, every GEMM as three bf16 products of hi / lo operand halves on
# the tile kernel — ops.gemm_split3, csrc/split3.hip: fp32-grade products at a third of the bf16 rate instead of 1/80)
_DT = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": torch.float32, torch.bfloat16: torch.bfloat16,
       torch.float32: torch.float32}


# ----------------------------------------------------------------- modules --
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class VisionTransformer(nn.Module):
    """`apply_head=False` reproduces upstream DINO, whose forward returns
    norm(x)[:, 0] and never calls `.head` ([recall]); the factory
    passes `apply_head=True` when it installs a classifier."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=0, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, eps=1e-6, apply_head=False,
                 compute_dtype="bf16", residual_dtype="fp32", cls_only_last_block=False, **_ignored):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.apply_head = apply_head
        # Opt-in (default off; bench.py reports it as a separate extra, never as `value`): the model's output is
        # norm(x)[:, 0], so in the LAST block only the CLS row of the attention output, of proj and of the MLP reaches it —
        # and only the CLS row carries a gradient back.  With the flag the last block computes exactly that row (k, v
        # still come from all tokens): the same logits, loss and gradients as the full computation (the reference
        # computes the other 196 rows and discards them), ~6 % fewer FLOPs per step.  tests/test_vit_gpu.py compares.
        self.cls_only_last_block = bool(cls_only_last_block)
        self.compute_dtype = _DT[compute_dtype]
        self.split3 = compute_dtype == "bf16x3"
        # the residual stream between the blocks: "fp32" (default: the reference trains in fp32, and the fp32 stream ends a
        # 50-step run within 0.06-0.08 % of the oracle loss, tests/test_training_curve_gpu.py), "bf16" (what bench.py times:
        # 1.1-2.2 % on the same test), or "auto" = follow the compute dtype
        self.residual_dtype = self.compute_dtype if residual_dtype == "auto" else _DT[residual_dtype]
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, eps)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        _trunc_normal_(self.pos_embed)
        _trunc_normal_(self.cls_token)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        self._engine: Optional[VitEngine] = None

    def engine(self) -> "VitEngine":
        if self._engine is None or not self._engine.is_current():
            self._engine = VitEngine(self)
        return self._engine

    def forward(self, x):
        if not x.is_cuda:
            raise VitmiError("vit_torch_amd models run on an MI355X (HIP) device; got a CPU tensor "
                             "and there is no CPU fallback")
        eng = self.engine()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in eng.pack.params)
        if need_grad:
            return _EngineFn.apply(eng, x, *eng.pack.params)
        return eng.forward(x, save=False)


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, *params):
        ctx.eng = eng
        return eng.forward(x, save=True)

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        held = eng.pack.begin_backward()        # .grad tensors that alias the flat buffer: accumulate, do not overwrite
        eng.backward(dout)
        eng.pack.end_backward(held)
        grads = []
        for p, gv in zip(eng.pack.params, eng.pack.fresh_grad_views()):
            if not p.requires_grad:
                grads.append(None)
            elif p.grad is not None and p.grad.data_ptr() == gv.data_ptr():
                grads.append(None)      # .grad already IS this buffer (zero_grad(set_to_none=False))
            else:
                grads.append(gv)
        return (None, None, *grads)


# ------------------------------------------------------------------ engine --
def _head_layers(head) -> Optional[List[tuple]]:
    """[(Linear, gelu_after)] for Identity / Linear / Sequential(Linear[,GELU]...) heads
    (/root/reference/models/vision_all.py:299-320); None if the head is something else."""
    if isinstance(head, nn.Identity):
        return []
    if isinstance(head, nn.Linear):
        return [(head, False)]
    if isinstance(head, nn.Sequential):
        out = []
        mods = list(head)
        i = 0
        while i < len(mods):
            if not isinstance(mods[i], nn.Linear):
                return None
            gelu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU)
            out.append((mods[i], gelu))
            i += 2 if gelu else 1
        return out
    return None


def _timing_events(eng):
    """Two timing events for one launch of the instrumented step.  `eng.profile_external`: the step is being CAPTURED into a
    HIP graph — `external` events become event-record nodes of the graph, so the per-launch durations are those of the
    replayed graph (the launch form bench.py times), not of an eager step."""
    ext = bool(getattr(eng, "profile_external", False))
    return (torch.cuda.Event(enable_timing=True, external=ext), torch.cuda.Event(enable_timing=True, external=ext))


def _rows_padded(*tensors) -> bool:
    """True when every given [M, .] tensor's storage reaches ceil256(M) rows at its row stride."""
    for t in tensors:
        if t is None or t.dim() != 2:
            continue
        rows = (t.shape[0] + 255) // 256 * 256
        need = (t.storage_offset() + (rows - 1) * t.stride(0) + t.shape[1]) * t.element_size()
        if t.untyped_storage().nbytes() < need:
            return False
    return True


def _engine_gemm_split3(eng, A, B, C, **k):
    """engine_gemm in the "bf16x3" mode: fp32 operands -> three-part bf16 images -> ONE bf16 product over 3K on the tile
    kernels, fp32 epilogue (ops.gemm_split3).  Profiled launches count the algorithmic 2 M N K FLOPs (the mode's own
    roofline is a third of the bf16 peak)."""
    if eng.reducer is not None:
        k.setdefault("launch_flags", eng.reducer.launch_flags())
    if k.get("C2") is not None and k.get("epilogue") == EPI_RESIDUAL:
        # CaiT's LayerScale residual x + gamma * f(x) with the branch output f kept for d gamma (models/cait.py:143-150): the
        # tile kernel writes that second output in the OPERAND dtype (bf16 here), so the mode takes three steps, each the
        # reference's own rounding: f = A W^T + b (three-product GEMM, fp32), out = gamma * f, out += x
        if k.get("rowscale") is not None or not (C.is_contiguous() and k["R"].is_contiguous()):
            raise VitmiError("compute_dtype='bf16x3': LayerScale residual with a DropPath row scale / strided rows is not built")
        f, R_, gamma = k.pop("C2"), k.pop("R"), k.pop("gamma", None)
        k.pop("epilogue")
        _engine_gemm_split3(eng, A, B, f, **k)
        ops.scale_cast(f, C, gamma, M=C.shape[0], N=C.shape[1])
        ops.axpy(R_.reshape(-1), C.reshape(-1), 1.0)
        return C
    if eng.profile is None:
        return ops.gemm_split3(A, B, C, **k)
    akm, bkm = k.get("a_kmajor", True), k.get("b_kmajor", True)
    Kdim = A.shape[1] if akm else A.shape[0]
    e0, e1 = _timing_events(eng)
    e0.record()
    ops.gemm_split3(A, B, C, **k)
    e1.record()
    name = "gemm3_" + ("n" if akm else "t") + ("t" if bkm else "n")
    eng.profile.append((name, (C.shape[0], C.shape[1], Kdim), 2.0 * C.shape[0] * C.shape[1] * Kdim, e0, e1))
    return C


def engine_gemm(eng, A, B, C, **k):
    """ops.gemm with the engine's implementation switch; when eng.profile is a list, each
    launch is bracketed by HIP events on the launch stream (bench.py's roofline leg)."""
    # the block MLPs keep gelu'(pre) instead of pre in bf16 mode: the GELU epilogue has the exp
    # at hand, and the backward epilogue becomes a multiply (vitmi_gemm_desc.aux_is_derivative).
    # fp32 (parity) mode keeps the pre-activation, as the reference's autograd does.
    if getattr(eng, "split3", False) and A.dtype == torch.float32:
        return _engine_gemm_split3(eng, A, B, C, **k)
    if k.get("epilogue") in (EPI_BIAS_GELU, EPI_DGELU):
        k.setdefault("aux_deriv", eng.T == torch.bfloat16)
    if eng.reducer is not None:             # gradient buckets in flight: share the device with RCCL's kernels
        k.setdefault("launch_flags", eng.reducer.launch_flags())
    if getattr(eng, "pad_rows", False) and _rows_padded(C, k.get("C2"), k.get("R"), k.get("aux")):
        # every [M, .] activation of this engine comes from _alloc (rows rounded up to 256); the flag is only passed when the
        # storage behind each row-indexed output / side input really covers the padding (a future torch.empty
        # output would otherwise be a silent out-of-bounds write of up to 255 rows)
        k["launch_flags"] = k.get("launch_flags", 0) | LAUNCH_ROWS_PADDED
    if eng.profile is None:
        return ops.gemm(A, B, C, impl=eng.gemm_impl, **k)
    akm, bkm = k.get("a_kmajor", True), k.get("b_kmajor", True)
    Kdim = A.shape[1] if akm else A.shape[0]
    e0, e1 = _timing_events(eng)
    e0.record()
    ops.gemm(A, B, C, impl=eng.gemm_impl, **k)
    e1.record()
    name = "gemm_" + ("n" if akm else "t") + ("t" if bkm else "n")
    eng.profile.append((name, (C.shape[0], C.shape[1], Kdim), 2.0 * C.shape[0] * C.shape[1] * Kdim, e0, e1))
    return C


def engine_wgrad_pair(eng, dY0, X0, dW0, dY1, X1, dW1):
    """dW0 = dY0^T X0 and dW1 = dY1^T X1 (the proj and qkv weight gradients of an attention block) in one launch where
    the library pairs them (ops.gemm_pair / vitmi_gemm_pair), else as two GEMMs."""
    paired = (eng.gemm_impl == GEMM_AUTO and dY0.dtype == torch.bfloat16 and dW0.dtype == torch.float32
              and ops.gemm_pair_shares_a_launch(dW0.shape[0], dW0.shape[1], dW1.shape[0], dW1.shape[1], dY0.shape[0]))
    if not paired:
        eng._gemm(dY0, X0, dW0, a_kmajor=False, b_kmajor=False)
        eng._gemm(dY1, X1, dW1, a_kmajor=False, b_kmajor=False)
        return
    flags = eng.reducer.launch_flags() if eng.reducer is not None else 0
    if eng.profile is None:
        ops.gemm_pair(dY0, X0, dW0, dY1, X1, dW1, launch_flags=flags)
        return
    e0, e1 = _timing_events(eng)
    e0.record()
    ops.gemm_pair(dY0, X0, dW0, dY1, X1, dW1, launch_flags=flags)
    e1.record()
    K = dY0.shape[0]
    flops = 2.0 * K * (dW0.shape[0] * dW0.shape[1] + dW1.shape[0] * dW1.shape[1])
    eng.profile.append(("gemm_tn_pair", (dW0.shape[0] + dW1.shape[0], dW0.shape[1], K), flops, e0, e1))


def dgelu_gemm_with_bias_grad(eng, Gb, W2, dH, pre, bias_grad):
    """dH = (Gb @ W2) * gelu'(pre) and bias_grad = column sums of dH.  On the bf16 tile paths
    the sums ride on the GEMM epilogue as per-128-row partials (a few MB, folded by colsum);
    otherwise dH is summed by a separate pass.  Returns a thunk that performs the fold /
    pass (callers may run it on a side stream)."""
    M, Dh = dH.shape
    K = Gb.shape[1]
    fused = (dH.dtype == torch.bfloat16 and eng.gemm_impl == GEMM_AUTO
             and ops.gemm_uses_fast(M, Dh, K, b_kmajor=False, epilogue=EPI_DGELU, colsum_part=True))
    if fused:
        part = torch.empty(((M + 127) // 128, Dh), dtype=torch.float32, device=dH.device)
        eng._gemm(Gb, W2, dH, b_kmajor=False, epilogue=EPI_DGELU, aux=pre, colsum_part=part)
        return lambda: ops.colsum(part, bias_grad, fold=getattr(eng, "folds", None))
    eng._gemm(Gb, W2, dH, b_kmajor=False, epilogue=EPI_DGELU, aux=pre)
    return lambda: ops.colsum(dH, bias_grad)


class VitEngine:
    """Executes VisionTransformer forward / backward as a fixed kernel sequence.

    Activations of type T (compute dtype: bf16, or fp32 in parity mode), residual
    stream of type R; every residual update writes a NEW buffer, so the inputs of
    every LayerNorm are kept for backward without an extra copy.
    """

    def __init__(self, model: VisionTransformer):
        self.model = model
        dev = model.pos_embed.device
        if dev.type != "cuda":
            raise VitmiError("move the model to the GPU before the first forward")
        self.T = model.compute_dtype
        self.R = model.residual_dtype
        self.split3 = bool(getattr(model, "split3", False))
        if self.T == torch.float32 and self.R != torch.float32:
            raise VitmiError("fp32 compute needs an fp32 residual stream")
        self.head = _head_layers(model.head) if model.apply_head else []
        if self.head is None:
            raise VitmiError("head must be Identity, Linear or Sequential(Linear[, GELU], ...)")
        named = [(n, p) for n, p in model.named_parameters()]
        self.pack = ParamPack(named, dev, shadow=self.T == torch.bfloat16)
        self.saved = None
        self.gemm_impl = GEMM_AUTO
        self.reducer = None          # ddp.GradReducer: told when a section's grads are final
        self.profile = None          # list of (name, flops, start_event, end_event) when profiling
        # (round 1 ran the weight-gradient GEMMs on a second HIP stream; A/B on one box,
        # profiles/r02_ab_overlap*: 38.47-38.67 ms/step without it, 38.55-39.26 with it — a
        # 256x256-tile GEMM workgroup owns its CU's whole register file and 128 KiB of LDS, so
        # two GEMM kernels only take CUs from each other.  Removed.)
        self.fused_bias_grads = os.environ.get("VITMI_FUSED_BIAS_GRADS", "1") != "0"
        # the ~50 small folds of a backward pass (LayerNorm dgamma | dbeta | bias sums, fc1 / qkv bias partials) run as
        # ONE launch per flush instead of one each (ops.FoldQueue; VITMI_DEFER_FOLDS=0: fold at once, for A/B)
        self.folds = ops.FoldQueue() if os.environ.get("VITMI_DEFER_FOLDS", "1") != "0" else None
        # Row padding (round 4): M = B * N is a multiple of the 256-row GEMM tile only for special batch sizes (197 B: B % 256
        # == 0).  Every [M, .] activation is allocated to the next multiple of 256 rows and used through its [:M] view; the
        # GEMM calls carry VITMI_LAUNCH_ROWS_PADDED, so a ragged M runs on the 256x256 tile kernel (last row tile: A's last row
        # repeated, surplus output rows into the padding) instead of the slower 256x128 ragged form.  VITMI_PAD_ROWS=0: off.
        self.pad_rows = self.T == torch.bfloat16 and os.environ.get("VITMI_PAD_ROWS", "1") != "0"
        self.cls_last = bool(getattr(model, "cls_only_last_block", False))
        if self.cls_last and model.embed_dim // model.blocks[0].attn.num_heads > 64:
            # the one-query form runs on the class-attention kernels (cait_ops.hip): hd <= 64 (and N <= 256, checked per input)
            raise VitmiError("cls_only_last_block needs head_dim <= 64 (the class-attention kernels' limit); got "
                             f"{model.embed_dim // model.blocks[0].attn.num_heads}: build the model without the option")

    def _alloc(self, rows, cols, dt, dev, zero=False):
        r = (rows + 255) // 256 * 256 if self.pad_rows else rows
        t = (torch.zeros if zero else torch.empty)((r, cols), dtype=dt, device=dev)
        return t[:rows]

    def is_current(self) -> bool:
        m = self.model
        return (self.pack.is_current() and m.compute_dtype == self.T and m.residual_dtype == self.R
                and bool(getattr(m, "split3", False)) == self.split3
                and len(self.pack.params) == sum(1 for _ in m.parameters())
                and bool(getattr(m, "cls_only_last_block", False)) == self.cls_last)

    # -- helpers -------------------------------------------------------------
    def _w(self, p):
        return self.pack.w(p)

    def _gemm(self, A, B, C, **k):
        return engine_gemm(self, A, B, C, **k)

    def _ready(self, *mods_or_params):
        if self.reducer is None:
            return
        if self.folds is not None:
            self.folds.flush()              # the section's bias / LayerNorm gradients must be final before its bucket leaves
        ps = []
        for o in mods_or_params:
            ps.extend(o.parameters() if isinstance(o, nn.Module) else [o])
        self.reducer.section_ready(ps)

    def _pos_for(self, gh, gw):
        """pos_embed at the input's patch grid: as stored, or the bicubic resize upstream DINO
        applies ([recall]; oracle/vit_ref.py:110-127) through the tap-table kernel
        (posembed.py, vitmi_pos_resample); returns (pos [N,D] fp32, tables or None)."""
        m = self.model
        pos = self.pack.f32(m.pos_embed)
        n_stored = pos.shape[1] - 1
        if gh * gw == n_stored and gh == gw:
            return pos.reshape(-1, pos.shape[-1]), None
        tabs = tables_for(n_stored, gh, gw, pos.device)
        return ops.pos_resample(pos.reshape(-1, pos.shape[-1]), tabs.fwd), tabs

    # -- forward -------------------------------------------------------------
    def forward(self, x, save: bool):
        m, T, R = self.model, self.T, self.R
        dev = x.device
        pre = x if isinstance(x, PatchRows) else None          # device input pipeline: rows already gathered
        if pre is None:
            x = x.float() if x.dtype != torch.float32 else x
            B, Cin, Himg, Wimg = x.shape
        else:
            B, Cin, Himg, Wimg = pre.B, pre.C, pre.H, pre.W
        p = m.patch_embed.patch_size
        conv = m.patch_embed.proj
        if Cin != conv.in_channels:
            raise VitmiError(f"input has {Cin} channels, patch_embed.proj expects {conv.in_channels}")
        gh, gw = Himg // p, Wimg // p
        N = 1 + gh * gw
        M = B * N
        D = m.embed_dim
        H = m.blocks[0].attn.num_heads
        hd = D // H
        Kp = Cin * p * p
        if self.cls_last and N > 256:
            raise VitmiError(f"cls_only_last_block needs at most 256 tokens per image (the class-attention kernels' limit); this "
                             f"input has {N}: build the model without the option")
        self.pack.refresh_shadow()

        def new(rows, cols, dt):
            return self._alloc(rows, cols, dt, dev)

        if pre is None:
            patches = new(M, Kp, T)
            ops.patchify(x, patches, p, cls_rows=1)
        else:
            if pre.p != p or pre.cls_rows != 1 or pre.rows.dtype != T or tuple(pre.rows.shape) != (M, Kp):
                raise VitmiError(f"PatchRows (p={pre.p}, cls_rows={pre.cls_rows}, {pre.rows.dtype}, {tuple(pre.rows.shape)}) "
                                 f"does not fit this model (p={p}, cls_rows=1, {T}, {(M, Kp)})")
            patches = pre.rows
        pos, pos_tabs = self._pos_for(gh, gw)
        X = new(M, D, R)
        self._gemm(patches, self._w(conv.weight).view(D, Kp), X, epilogue=EPI_PATCH_POS,
                   bias=self.pack.f32(conv.bias) if conv.bias is not None else None,
                   pos=pos, n_tok=N, cls=self.pack.f32(m.cls_token).view(-1))
        blocks = []
        last_cls = None
        for bi_, blk in enumerate(m.blocks):
            if self.cls_last and bi_ == len(m.blocks) - 1:
                X, last_cls = self._last_block_cls_fwd(blk, X, B, N, M, D, H, hd, save, new, dev)
                break
            a, mlp = blk.attn, blk.mlp
            ln1 = new(M, D, T)
            mean1 = torch.empty(M, dtype=torch.float32, device=dev)
            rstd1 = torch.empty(M, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(X, self.pack.f32(blk.norm1.weight), self.pack.f32(blk.norm1.bias), ln1,
                              mean1, rstd1, blk.norm1.eps, M=M, D=D)
            qkv = new(M, 3 * D, T)
            self._gemm(ln1, self._w(a.qkv.weight), qkv,
                       bias=self.pack.f32(a.qkv.bias) if a.qkv.bias is not None else None)
            O = new(M, D, T)
            lse = torch.empty(B * H * N, dtype=torch.float32, device=dev)
            ops.attn_fwd(qkv, O, lse, B, N, H, hd, a.scale)
            X1 = new(M, D, R)
            self._gemm(O, self._w(a.proj.weight), X1, epilogue=EPI_RESIDUAL,
                       bias=self.pack.f32(a.proj.bias), R=X)
            ln2 = new(M, D, T)
            mean2 = torch.empty(M, dtype=torch.float32, device=dev)
            rstd2 = torch.empty(M, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(X1, self.pack.f32(blk.norm2.weight), self.pack.f32(blk.norm2.bias), ln2,
                              mean2, rstd2, blk.norm2.eps, M=M, D=D)
            Dh = mlp.fc1.out_features
            pre = new(M, Dh, T) if save else None       # what the backward needs of fc1's output (engine_gemm)
            hid = new(M, Dh, T)
            self._gemm(ln2, self._w(mlp.fc1.weight), hid, epilogue=EPI_BIAS_GELU,
                       bias=self.pack.f32(mlp.fc1.bias), C2=pre)
            X2 = new(M, D, R)
            self._gemm(hid, self._w(mlp.fc2.weight), X2, epilogue=EPI_RESIDUAL,
                       bias=self.pack.f32(mlp.fc2.bias), R=X1)
            if save:
                blocks.append((X, ln1, mean1, rstd1, qkv, O, lse, X1, ln2, mean2, rstd2, pre, hid))
            X = X2
        feat = torch.empty((B, D), dtype=torch.float32, device=dev)
        meanf = torch.empty(B, dtype=torch.float32, device=dev)
        rstdf = torch.empty(B, dtype=torch.float32, device=dev)
        xf_stride = D if last_cls is not None else N * D       # the CLS-only last block leaves a [B, D] stream
        ops.layernorm_fwd(X, self.pack.f32(m.norm.weight), self.pack.f32(m.norm.bias), feat, meanf,
                          rstdf, m.norm.eps, M=B, D=D, x_stride=xf_stride, y_stride=D)
        # classifier head: tiny fp32 GEMMs on the generic MFMA kernel
        acts = [feat]
        pres = []
        cur = feat
        for lin, gelu in self.head:
            out = torch.empty((B, lin.out_features), dtype=torch.float32, device=dev)
            bias = self.pack.f32(lin.bias) if lin.bias is not None else None
            if gelu:
                pre_h = torch.empty_like(out)
                ops.gemm(cur, self.pack.f32(lin.weight), out, epilogue=EPI_BIAS_GELU, bias=bias, C2=pre_h)
                pres.append(pre_h)
            else:
                ops.gemm(cur, self.pack.f32(lin.weight), out, bias=bias)
                pres.append(None)
            acts.append(out)
            cur = out
        if save:
            self.saved = dict(B=B, N=N, M=M, D=D, H=H, hd=hd, Kp=Kp, patches=patches, blocks=blocks,
                              Xf=X, meanf=meanf, rstdf=rstdf, acts=acts, pres=pres,
                              pos_tabs=pos_tabs, last_cls=last_cls)
        return cur

    # -- the last block on the CLS row only (cls_only_last_block) ------------------
    def _last_block_cls_fwd(self, blk, X, B, N, M, D, H, hd, save, new, dev):
        """x_cls' = block(x)[:, 0]: LayerNorm and the k / v projections over all tokens, one query per image (the
        class-attention kernels of cait_ops.hip: softmax((q k^T) scale) v with q from token 0), proj / residual / MLP on B rows.
        Returns the new CLS stream [B, D] (dtype R) and what the backward needs."""
        T, R, pk = self.T, self.R, self.pack
        a, mlp = blk.attn, blk.mlp
        f32 = torch.float32
        ln1 = new(M, D, T)
        mean1, rstd1 = torch.empty(M, dtype=f32, device=dev), torch.empty(M, dtype=f32, device=dev)
        ops.layernorm_fwd(X, pk.f32(blk.norm1.weight), pk.f32(blk.norm1.bias), ln1, mean1, rstd1, blk.norm1.eps, M=M, D=D)
        Wqkv, bqkv = self._w(a.qkv.weight), (pk.f32(a.qkv.bias) if a.qkv.bias is not None else None)
        kv = new(M, 2 * D, T)
        self._gemm(ln1, Wqkv[D:], kv, bias=bqkv[D:] if bqkv is not None else None)
        ln1_cls = ln1.view(B, N * D)[:, :D]                     # strided CLS rows
        q = torch.empty((B, D), dtype=T, device=dev)
        ops.gemm(ln1_cls, Wqkv[:D], q, bias=bqkv[:D] if bqkv is not None else None)
        oc = torch.empty((B, D), dtype=T, device=dev)
        psave = torch.empty(B * H * N, dtype=f32, device=dev)
        ops.class_attn_fwd(q, kv, kv[:, D:], 2 * D, oc, psave, B, H, N, hd, a.scale)
        Xc = X.view(B, N * D)[:, :D]                            # the residual stream's CLS rows (strided)
        X1 = torch.empty((B, D), dtype=R, device=dev)
        ops.gemm(oc, self._w(a.proj.weight), X1, epilogue=EPI_RESIDUAL, bias=pk.f32(a.proj.bias), R=Xc)
        ln2 = torch.empty((B, D), dtype=T, device=dev)
        mean2, rstd2 = torch.empty(B, dtype=f32, device=dev), torch.empty(B, dtype=f32, device=dev)
        ops.layernorm_fwd(X1, pk.f32(blk.norm2.weight), pk.f32(blk.norm2.bias), ln2, mean2, rstd2, blk.norm2.eps, M=B, D=D)
        Dh = mlp.fc1.out_features
        pre = torch.empty((B, Dh), dtype=T, device=dev) if save else None
        hid = torch.empty((B, Dh), dtype=T, device=dev)
        ops.gemm(ln2, self._w(mlp.fc1.weight), hid, epilogue=EPI_BIAS_GELU, bias=pk.f32(mlp.fc1.bias), C2=pre,
                 aux_deriv=T == torch.bfloat16)
        X2 = torch.empty((B, D), dtype=R, device=dev)
        ops.gemm(hid, self._w(mlp.fc2.weight), X2, epilogue=EPI_RESIDUAL, bias=pk.f32(mlp.fc2.bias), R=X1)
        saved = (X, ln1, mean1, rstd1, kv, q, oc, psave, X1, ln2, mean2, rstd2, pre, hid) if save else None
        return X2, saved

    def _last_block_cls_bwd(self, blk, sv, Gc, B, N, M, D, H, hd, dev, prev_fc2_bias):
        """Backward of _last_block_cls_fwd.  Gc [B, D] (dtype R): gradient of the block's CLS output.  Returns G [M, D]
        (dtype R, padded rows): the gradient of the residual stream entering the block, and its operand copy Gb."""
        T, R, pk = self.T, self.R, self.pack
        a, mlp = blk.attn, blk.mlp
        f32 = torch.float32
        X, ln1, mean1, rstd1, kv, q, oc, psave, X1, ln2, mean2, rstd2, pre, hid = sv
        Dh = mlp.fc1.out_features

        def cast(t):
            if t.dtype == T:
                return t
            o = torch.empty(t.shape, dtype=T, device=dev)
            ops.cast(t.contiguous(), o)
            return o

        Gcb = cast(Gc)
        # MLP branch on B rows
        dH = torch.empty((B, Dh), dtype=T, device=dev)
        ops.gemm(Gcb, self._w(mlp.fc2.weight), dH, b_kmajor=False, epilogue=EPI_DGELU, aux=pre, aux_deriv=T == torch.bfloat16)
        ops.gemm(Gcb, hid, pk.g(mlp.fc2.weight), a_kmajor=False, b_kmajor=False)
        ops.colsum(Gcb, pk.g(mlp.fc2.bias))
        ops.gemm(dH, ln2, pk.g(mlp.fc1.weight), a_kmajor=False, b_kmajor=False)
        ops.colsum(dH, pk.g(mlp.fc1.bias))
        dln2 = torch.empty((B, D), dtype=T, device=dev)
        ops.gemm(dH, self._w(mlp.fc1.weight), dln2, b_kmajor=False)
        G1 = torch.empty((B, D), dtype=R, device=dev)            # gradient of X1 = Gc + LN2 backward
        G1b = torch.empty((B, D), dtype=T, device=dev) if T != R else None
        ops.layernorm_bwd(dln2, X1, mean2, rstd2, pk.f32(blk.norm2.weight), Gc.contiguous(), G1, G1b,
                          pk.g(blk.norm2.weight), pk.g(blk.norm2.bias), gsum=pk.g(a.proj.bias), M=B, D=D)
        G1o = G1 if G1b is None else G1b
        # attention branch: one query per image
        doc = torch.empty((B, D), dtype=T, device=dev)
        ops.gemm(G1o, self._w(a.proj.weight), doc, b_kmajor=False)
        ops.gemm(G1o, oc, pk.g(a.proj.weight), a_kmajor=False, b_kmajor=False)
        dq = torch.empty((B, D), dtype=T, device=dev)
        dkv = self._alloc(M, 2 * D, T, dev)
        ops.class_attn_bwd(q, kv, kv[:, D:], 2 * D, doc, psave, dq, dkv, dkv[:, D:], 2 * D, B, H, N, hd, a.scale)
        Wqkv = self._w(a.qkv.weight)
        gW = pk.g(a.qkv.weight)
        ln1_cls = ln1.view(B, N * D)[:, :D]
        ops.gemm(dkv, ln1, gW[D:], a_kmajor=False, b_kmajor=False)
        ops.gemm(dq, ln1_cls, gW[:D], a_kmajor=False, b_kmajor=False)
        if a.qkv.bias is not None:
            gb = pk.g(a.qkv.bias)
            ops.colsum(dkv, gb[D:])
            ops.colsum(dq, gb[:D])
        # d ln1 = dkv Wkv (+ dq Wq on the CLS rows), accumulated in fp32
        dln1 = self._alloc(M, D, f32, dev)
        self._gemm(dkv, Wqkv[D:], dln1, b_kmajor=False)
        ops.gemm(dq, Wqkv[:D], dln1.view(B, N * D)[:, :D], b_kmajor=False, accumulate=True)
        # residual gradient entering the block: zero except the CLS rows (= G1), plus LN1 backward
        G = self._alloc(M, D, R, dev, zero=True)
        ops.scale_cast(G1, G.view(B, N * D), M=B, N=D, ldo=N * D)
        Gb = None if T == R else self._alloc(M, D, T, dev)
        ops.layernorm_bwd(dln1, X, mean1, rstd1, pk.f32(blk.norm1.weight), G, G, Gb,
                          pk.g(blk.norm1.weight), pk.g(blk.norm1.bias),
                          gsum=pk.g(prev_fc2_bias) if prev_fc2_bias is not None else None, M=M, D=D, fold=self.folds)
        return G, (G if Gb is None else Gb)

    # -- backward ------------------------------------------------------------
    def backward(self, dout):
        try:
            self._backward(dout)
        except BaseException:
            if self.folds is not None:
                self.folds.clear()
            if self.reducer is not None:
                self.reducer.abort()
            raise

    def _backward(self, dout):
        s = self.saved
        if s is None:
            raise VitmiError("backward called without a saved forward (or called twice)")
        self.saved = None
        m, T, R, pk = self.model, self.T, self.R, self.pack
        B, N, M, D, H, hd = s["B"], s["N"], s["M"], s["D"], s["H"], s["hd"]
        dev = dout.device
        d = dout.contiguous().float()

        def new(rows, cols, dt):
            return self._alloc(rows, cols, dt, dev)

        # ---- classifier head (fp32, generic MFMA kernel) ----
        # z_i = a_i W_i^T + b_i ; a_{i+1} = gelu(z_i) or z_i.  `d` is dL/dz_i on entry;
        # the inner layer's gelu' is applied by the DGELU epilogue of this layer's dX GEMM.
        acts, pres = s["acts"], s["pres"]
        if self.head and self.head[-1][1]:
            raise VitmiError("a head ending in GELU is not supported")
        for li in range(len(self.head) - 1, -1, -1):
            lin, _ = self.head[li]
            ops.gemm(d, acts[li], pk.g(lin.weight), a_kmajor=False, b_kmajor=False)
            if lin.bias is not None:
                ops.colsum(d, pk.g(lin.bias))
            dx = torch.empty((B, lin.in_features), dtype=torch.float32, device=dev)
            if li > 0 and self.head[li - 1][1]:
                ops.gemm(d, pk.f32(lin.weight), dx, b_kmajor=False, epilogue=EPI_DGELU, aux=pres[li - 1])
            else:
                ops.gemm(d, pk.f32(lin.weight), dx, b_kmajor=False)
            d = dx
        dfeat = d

        # ---- final LayerNorm on the CLS rows -> residual-stream gradient G ----
        blocks_list = list(m.blocks)
        if s.get("last_cls") is not None:
            # CLS-only last block: the stream after it is [B, D]; its backward rebuilds the full-width gradient
            Gc = torch.empty((B, D), dtype=R, device=dev)
            ops.layernorm_bwd(dfeat, s["Xf"], s["meanf"], s["rstdf"], pk.f32(m.norm.weight), None, Gc, None,
                              pk.g(m.norm.weight), pk.g(m.norm.bias), M=B, D=D, dy_stride=D, x_stride=D, g_stride=D,
                              fold=self.folds)
            self._ready(m.norm, *([m.head] if self.head else []))
            last = blocks_list.pop()
            prev = blocks_list[-1].mlp.fc2.bias if blocks_list else None
            G, Gb = self._last_block_cls_bwd(last, s["last_cls"], Gc, B, N, M, D, H, hd, dev, prev)
            s["last_cls"] = None
            self._ready(last)
        else:
            G = self._alloc(M, D, R, dev, zero=True)
            # gsum of an LN backward = column sum of the gradient it leaves in G = the bias
            # gradient of the Linear (fc2 / proj) that wrote that residual position
            last_fc2_bias = m.blocks[-1].mlp.fc2.bias
            ops.layernorm_bwd(dfeat, s["Xf"], s["meanf"], s["rstdf"], pk.f32(m.norm.weight), None, G, None,
                              pk.g(m.norm.weight), pk.g(m.norm.bias), gsum=pk.g(last_fc2_bias), M=B, D=D,
                              dy_stride=D, x_stride=N * D, g_stride=N * D, fold=self.folds)
            self._ready(m.norm, *([m.head] if self.head else []))
            if T == R:
                Gb = G                      # GEMM operand and residual gradient share one buffer
            else:
                Gb = new(M, D, T)
                ops.cast(G, Gb)
        gb_out = None if T == R else Gb

        fused_bias = self.fused_bias_grads and T == torch.bfloat16 and self.gemm_impl == GEMM_AUTO

        saved_blocks = s["blocks"]
        for bi in range(len(blocks_list) - 1, -1, -1):
            blk = blocks_list[bi]
            sv = saved_blocks.pop()      # release each block's activations as we go
            X, ln1, mean1, rstd1, qkv, O, lse, X1, ln2, mean2, rstd2, pre, hid = sv
            del sv
            a, mlp = blk.attn, blk.mlp
            Dh = mlp.fc1.out_features
            # MLP branch
            dH = new(M, Dh, T)
            # bias gradients ride on the kernels that produce dH / dqkv (per-row-block column
            # sums, folded by a tiny colsum) whenever those kernels are the bf16 fast ones
            dH_part = None
            if fused_bias and ops.gemm_uses_fast(M, Dh, D, b_kmajor=False, epilogue=EPI_DGELU, colsum_part=True):
                dH_part = torch.empty(((M + 127) // 128, Dh), dtype=torch.float32, device=dev)
            self._gemm(Gb, self._w(mlp.fc2.weight), dH, b_kmajor=False, epilogue=EPI_DGELU, aux=pre,
                       **({"colsum_part": dH_part} if dH_part is not None else {}))
            self._gemm(Gb, hid, pk.g(mlp.fc2.weight), a_kmajor=False, b_kmajor=False)
            self._gemm(dH, ln2, pk.g(mlp.fc1.weight), a_kmajor=False, b_kmajor=False)
            ops.colsum(dH_part if dH_part is not None else dH, pk.g(mlp.fc1.bias), fold=self.folds)
            dln2 = new(M, D, T)
            self._gemm(dH, self._w(mlp.fc1.weight), dln2, b_kmajor=False)
            ops.layernorm_bwd(dln2, X1, mean2, rstd2, pk.f32(blk.norm2.weight), G, G, gb_out,
                              pk.g(blk.norm2.weight), pk.g(blk.norm2.bias), gsum=pk.g(a.proj.bias),
                              M=M, D=D, fold=self.folds)
            # attention branch
            dO = new(M, D, T)
            self._gemm(Gb, self._w(a.proj.weight), dO, b_kmajor=False)
            dqkv = new(M, 3 * D, T)
            dqkv_part = None
            if fused_bias and a.qkv.bias is not None:
                dqkv_part = torch.empty((ops.attn_bwd_dbias_rows(B, N), 3 * D), dtype=torch.float32, device=dev)
            ops.attn_bwd(qkv, O, dO, lse, dqkv, B, N, H, hd, a.scale, dbias_part=dqkv_part,
                         launch_flags=self.reducer.launch_flags() if self.reducer is not None else 0)
            # the proj and qkv weight gradients share one split-K launch (Gb still holds this block's G' here: the
            # LayerNorm backward below is what overwrites it)
            engine_wgrad_pair(self, Gb, O, pk.g(a.proj.weight), dqkv, ln1, pk.g(a.qkv.weight))
            if a.qkv.bias is not None:
                ops.colsum(dqkv_part if dqkv_part is not None else dqkv, pk.g(a.qkv.bias), fold=self.folds)
            dln1 = new(M, D, T)
            self._gemm(dqkv, self._w(a.qkv.weight), dln1, b_kmajor=False)
            # the gradient this leaves in G flows into the previous block's fc2 output
            prev_fc2_bias = blocks_list[bi - 1].mlp.fc2.bias if bi > 0 else None
            ops.layernorm_bwd(dln1, X, mean1, rstd1, pk.f32(blk.norm1.weight), G, G, gb_out,
                              pk.g(blk.norm1.weight), pk.g(blk.norm1.bias),
                              gsum=pk.g(prev_fc2_bias) if prev_fc2_bias is not None else None, M=M, D=D,
                              fold=self.folds)
            self._ready(blk)

        # ---- embeddings ----
        conv = m.patch_embed.proj
        Kp = s["Kp"]
        dpos = torch.empty(N * D, dtype=torch.float32, device=dev)
        ops.colsum(G, dpos, M=B, N=N * D, ld=N * D)          # sum over the batch
        ops.cast(dpos[:D], pk.g(m.cls_token).view(-1))       # d cls = d pos[0]
        if s["pos_tabs"] is None:
            ops.cast(dpos, pk.g(m.pos_embed).view(-1))
        else:   # through the bicubic resize: the transposed tap table
            ops.pos_resample(dpos.view(N, D), s["pos_tabs"].bwd, pk.g(m.pos_embed).view(-1, D))
        self._gemm(Gb, s["patches"], pk.g(conv.weight).view(D, Kp), a_kmajor=False, b_kmajor=False)
        if conv.bias is not None:
            ops.colsum(dpos[D:].view(N - 1, D), pk.g(conv.bias))  # CLS rows carry no conv bias
        self._ready(m.cls_token, m.pos_embed, m.patch_embed)
        if self.folds is not None:
            self.folds.flush()
        if self.reducer is not None:
            self.reducer.finish()
