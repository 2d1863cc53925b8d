"""CLIP dual encoder for the embedding stage of the hot path.  On the GPU inference path every block
runs on the HIP library: tower GEMMs through lemon_linear_f32 (hipBLASLt with bias / SiLU / residual
epilogues), attention through lemon_attention_f32; PyTorch supplies LayerNorm, the patch convolution and
the tensors.  The CPU / autograd path (parity oracle for the encoder) is plain PyTorch.

Mirrors the model surface run_lemon.py uses:
  algorithm_class_from_scratch(name, text_base_name, img_base, return_tokenizer)   lib/models/utils.py:64-105
  model.encode_text(input_ids, attention_mask) / model.encode_image(pixel_values)  lib/models/downstream_models.py:30-41
The architecture is the one HF `CLIPModel` / the in-tree CheXzero copy implement
(lib/models/chexzero_clip.py:177-260,263-392): pre-LN transformer blocks, QuickGELU
(`x * sigmoid(1.702 x)`, :186-188), causal text mask (:348-354), EOT pooling by argmax of the token
ids (:363-376), bias-free projections.  Weights load from a LOCAL HF checkpoint directory
(config.json + model.safetensors | pytorch_model.bin); there is no network in this environment,
so loading by hub name is refused with a clear message.

MI355X-first choices: fused QKV projection (one [3W,W] GEMM instead of three), a fused attention kernel
instead of a materialised attention matrix, text batches truncated to the longest prompt (exact, because the mask
is causal and pooling reads the EOT position: SURVEY 3.2), distinct prompts embedded once and
gathered (classification datasets have C distinct prompts for N samples).
"""
import json
import math
import os
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class TowerConfig:
    width: int
    layers: int
    heads: int
    mlp: int


@dataclass
class ClipConfig:
    embed_dim: int = 512
    image_size: int = 224
    patch_size: int = 32
    vision: TowerConfig = field(default_factory=lambda: TowerConfig(768, 12, 12, 3072))
    text: TowerConfig = field(default_factory=lambda: TowerConfig(512, 12, 8, 2048))
    vocab_size: int = 49408
    context_length: int = 77
    eos_token_id: int = 49407
    layer_norm_eps: float = 1e-5

    @staticmethod
    def named(name):
        """The three architectures BASELINE.json's configs name (the reference hard-codes B/32,
        run_lemon.py:113; selecting another one is an explicit extension)."""
        name = name.lower().replace("openai/clip-", "").replace("_", "-")
        if name in ("vit-b-32", "vit-base-patch32", "b32"):
            return ClipConfig()
        if name in ("vit-b-16", "vit-base-patch16", "b16"):
            return ClipConfig(patch_size=16)
        if name in ("vit-l-14", "vit-large-patch14", "l14"):
            return ClipConfig(embed_dim=768, patch_size=14, vision=TowerConfig(1024, 24, 16, 4096),
                              text=TowerConfig(768, 12, 12, 3072))
        if name in ("chexzero-scratch-256", "mimic-clip-from-scratch", "scratch-b16-256"):
            # lib/models/chexzero_clip.py:458-470 load_clip(): ViT-B/16 vision, 512-wide text, 768-d, context 256
            return ClipConfig(embed_dim=768, patch_size=16, context_length=256)
        if name in ("chexzero-scratch-77", "cc3m-clip-from-scratch", "scratch-b16-77"):
            return ClipConfig(embed_dim=768, patch_size=16, context_length=77)          # load_clip(context_length=77)
        if name in ("tiny", "test"):
            return ClipConfig(embed_dim=32, image_size=32, patch_size=8, vision=TowerConfig(48, 2, 4, 96),
                              text=TowerConfig(40, 2, 4, 80), vocab_size=300, context_length=16, eos_token_id=299)
        raise ValueError(f"unknown CLIP architecture {name!r}")

    @staticmethod
    def from_hf_dict(c):
        v, t = c["vision_config"], c["text_config"]
        return ClipConfig(
            embed_dim=c.get("projection_dim", 512), image_size=v.get("image_size", 224),
            patch_size=v.get("patch_size", 32),
            vision=TowerConfig(v.get("hidden_size", 768), v.get("num_hidden_layers", 12),
                               v.get("num_attention_heads", 12), v.get("intermediate_size", 3072)),
            text=TowerConfig(t.get("hidden_size", 512), t.get("num_hidden_layers", 12),
                             t.get("num_attention_heads", 8), t.get("intermediate_size", 2048)),
            vocab_size=t.get("vocab_size", 49408), context_length=t.get("max_position_embeddings", 77),
            eos_token_id=t.get("eos_token_id", 49407), layer_norm_eps=v.get("layer_norm_eps", 1e-5))

    @staticmethod
    def from_openai_state_dict(sd):
        """Architecture of an OpenAI-format CLIP state dict, read off the tensor shapes the way
        lib/models/chexzero_clip.py:411-441 (build_model) does.  ViT visual towers only."""
        if "visual.proj" not in sd:
            raise NotImplementedError("ResNet visual towers (ModifiedResNet, chexzero_clip.py:94-174) are not on LEMoN's path")
        vw = sd["visual.conv1.weight"].shape[0]
        patch = sd["visual.conv1.weight"].shape[-1]
        grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        tw = sd["ln_final.weight"].shape[0]
        tl = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
        vocab = sd["token_embedding.weight"].shape[0]
        return ClipConfig(embed_dim=sd["text_projection"].shape[1], image_size=patch * grid, patch_size=patch,
                          vision=TowerConfig(vw, vl, vw // 64, sd["visual.transformer.resblocks.0.mlp.c_fc.weight"].shape[0]),
                          text=TowerConfig(tw, tl, tw // 64, sd["transformer.resblocks.0.mlp.c_fc.weight"].shape[0]),
                          vocab_size=vocab, context_length=sd["positional_embedding"].shape[0], eos_token_id=vocab - 1)


def split_weight_cached(owner, name, w, ops, mode):
    """The weight `w` ([n, k] view of a parameter of `owner`) as the split GEMM operand of `mode` ([n, 6k] bf16 / [n, 3k] fp16 of
    w * wscale) and 1 / wscale, made once per weight version (inference: once) and kept on the module."""
    cache = owner.__dict__.setdefault("_split_cache", {})
    hit = cache.get((name, mode))
    if hit is None or hit[0] != (w.data_ptr(), w._version):
        wscale = ops.weight_scale_f16x3(w) if mode == "f16x3" else 1.0
        hit = ((w.data_ptr(), w._version), ops.split_operand(w.detach(), mode, weight=True, wscale=wscale), 1.0 / wscale)
        cache[(name, mode)] = hit
    return hit[1], hit[2]


class Block(nn.Module):
    def __init__(self, cfg: TowerConfig, eps, act="quick_gelu"):
        super().__init__()
        assert act in ("quick_gelu", "gelu")
        self.heads = cfg.heads
        # 'quick_gelu': z sigmoid(1.702 z) (CLIP, chexzero_clip.py:186-188); 'gelu': the exact one (timm's ViT blocks inside
        # open_clip's BiomedCLIP, lib/models/utils.py:72-78)
        self.act = act
        self.ln1 = nn.LayerNorm(cfg.width, eps=eps)
        self.qkv = nn.Linear(cfg.width, 3 * cfg.width)
        self.out = nn.Linear(cfg.width, cfg.width)
        self.ln2 = nn.LayerNorm(cfg.width, eps=eps)
        self.fc1 = nn.Linear(cfg.width, cfg.mlp)
        self.fc2 = nn.Linear(cfg.mlp, cfg.width)

    def forward(self, x, causal, rows=None, carry=None):
        """x [B,L,W] -> [B,L,W]; with rows = (batch_index, token_index) the block's output for those tokens
        only, [len(rows), W].  The towers read ONE token of the last block (CLS / EOT pooling): attention
        still sees every token's keys and values, but the output projection and the MLP -- 3/4 of the block's
        FLOPs, all row-wise -- are then evaluated for the pooled rows only.  Same function of the input.
        carry: what forward_chain of the block in front left behind (x as the hand-written GEMM's operand + row statistics), or None."""
        B, L, W = x.shape
        fused = x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
        if fused:
            # inference on the GPU: every GEMM of the block goes through lemon_linear_f32 (bias, QuickGELU
            # and the residual adds ride in the hipBLASLt epilogue), attention through lemon_attention_f32
            from . import ops
            mode = ops.gemm_mode()
            ops.select_attention_arithmetic(mode)
            if W % 4 == 0 and W <= 1024 and mode != "f32":
                return self._forward_split(x, causal, rows, ops, mode, carry)
            ln = lambda m, t: ops.layer_norm(t, m.weight, m.bias, m.eps) if t.shape[-1] % 4 == 0 else m(t)
            qkv = ops.linear(ln(self.ln1, x), self.qkv.weight, self.qkv.bias)
            if W == 64 * self.heads and L <= ops.ATTENTION_MAX_SEQ:
                a = ops.attention(qkv, self.heads, causal)
            else:
                a = self._sdpa(qkv, B, L, W, causal)
            if rows is not None:
                a, x = a[rows].contiguous(), x[rows].contiguous()
            x = ops.linear(a, self.out.weight, self.out.bias, residual=x)
            # QuickGELU(z) = silu(1.702 z) / 1.702: scale going in (alpha, bias), un-scale in fc2's alpha
            s, act = self._act_scale(ops)
            h = ops.linear(ln(self.ln2, x), self.fc1.weight, self._fc1_bias_scaled(s), act=act, alpha=s)
            return ops.linear(h, self.fc2.weight, self.fc2.bias, residual=x, alpha=1.0 / s)
        a = self._sdpa(self.qkv(self.ln1(x)), B, L, W, causal)
        if rows is not None:
            a, x = a[rows], x[rows]
        x = x + self.out(a)
        h = self.fc1(self.ln2(x))
        h = h * torch.sigmoid(1.702 * h) if self.act == "quick_gelu" else F.gelu(h)
        return x + self.fc2(h)

    def _act_scale(self, ops):
        """(s, epilogue name): fc1 runs as act(s (x W^T + b)) and fc2 with alpha / s -- QuickGELU(z) = silu(1.702 z) / 1.702;
        the exact GELU needs no scale."""
        return (ops.QUICK_GELU_SCALE, "silu") if self.act == "quick_gelu" else (1.0, "gelu")

    def _fc1_bias_scaled(self, s):
        """fc1.bias * s (the QuickGELU scale folded into the SiLU epilogue's input), made once per bias version."""
        b = self.fc1.bias
        cache = self.__dict__.setdefault("_split_cache", {})
        hit = cache.get("fc1_bias_scaled")
        if hit is None or hit[0] != (b.data_ptr(), b._version, s):
            hit = ((b.data_ptr(), b._version, s), (b.detach() * s).contiguous())
            cache["fc1_bias_scaled"] = hit
        return hit[1]

    def _mlp_hand(self, x, ops, W, mlp):
        """x + fc2(QuickGELU(fc1(LayerNorm(x)))) in the hand-written GEMM (gemm_f16x3.hip): LayerNorm writes the tile-major
        operand, fc1's epilogue applies bias + QuickGELU + the fp16 split and stores fc2's operand, fc2 adds bias and residual."""
        s, act = self._act_scale(ops)
        m = x.numel() // W
        w1, a1 = self._w_tiled("fc1", ops)
        w2, a2 = self._w_tiled("fc2", ops)
        at = ops.layer_norm_t(x, self.ln2.weight, self.ln2.bias, self.ln2.eps)
        ht = ops.linear_t(at, w1, m, mlp, W, self._fc1_bias_scaled(s), act=act, alpha=s * a1)
        return ops.linear_t(ht, w2, m, W, mlp, self.fc2.bias, residual=x, alpha=a2 / s, out_shape=x.shape)

    def _w_split(self, name, ops, mode):
        return split_weight_cached(self, name, getattr(self, name).weight, ops, mode)

    def _w_tiled(self, name, ops):
        """The layer's weight as the tile-major fp16 operand of lemon_linear_f16x3t and 1 / wscale, once per weight version."""
        w = getattr(self, name).weight
        cache = self.__dict__.setdefault("_split_cache", {})
        hit = cache.get((name, "tiled"))
        if hit is None or hit[0] != (w.data_ptr(), w._version):
            wscale = ops.weight_scale_f16x3(w)
            hit = ((w.data_ptr(), w._version), ops.pack_weight_t(w.detach(), wscale), 1.0 / wscale)
            cache[(name, "tiled")] = hit
        return hit[1], hit[2]

    def _forward_split(self, x, causal, rows, ops, mode, carry=None):
        """The fused inference path with the four GEMMs of the block (QKV, output projection, fc1, fc2) on the 16-bit matrix
        cores at fp32-equivalent accuracy (ops.linear_split: exact splits of both operands -- 3-way bf16, six cross products, or
        2-way fp16, three --, fp32 accumulate); LayerNorm and attention write the split operand directly, the MLP activations
        get one split pass."""
        B, L, W = x.shape
        mlp = self.fc1.weight.shape[0]
        hand = mode == "f16x3" and rows is None and ops.mlp_mode()
        if hand == "block" and ops.block_fused_supported(W, mlp, self.heads, L):
            # QKV and the output projection in the hand-written GEMM as well (LEMON_MLP=block, the default: see ops.mlp_mode):
            # LayerNorm and attention write its tile-major operands
            m = B * L
            wq, aq = self._w_tiled("qkv", ops)
            wo, ao = self._w_tiled("out", ops)
            qkv = ops.linear_t(ops.layer_norm_t(x, self.ln1.weight, self.ln1.bias, self.ln1.eps), wq, m, 3 * W, W, self.qkv.bias,
                               alpha=aq, out_shape=(B, L, 3 * W))
            x = ops.linear_t(ops.attention_t(qkv, self.heads, causal), wo, m, W, W, self.out.bias, residual=x, alpha=ao, out_shape=x.shape)
            return self._mlp_hand(x, ops, W, mlp)
        if rows is not None and mode == "f16x3" and ops.mlp_mode() == "block" and ops.block_fused_supported(W, mlp, self.heads, L):
            # the pooled-row block still needs every token's keys and values: its QKV GEMM (all rows) in the hand-written kernel
            # too, with the LayerNorm folded in when the block in front left its operand and statistics
            m = B * L
            if carry is not None and ops.ln_fold_enabled() and W % 32 == 0:
                wq, aq, csq, bq = self._w_tiled_ln("qkv", self.ln1, ops, 1.0)
                qkv = ops.linear_t_ln(carry[0], wq, m, 3 * W, W, bq, alpha=aq, out_shape=(B, L, 3 * W),
                                      row_aff=ops.ln_finalize(carry[1], m, W, self.ln1.eps), colsum=csq)
            else:
                wq, aq = self._w_tiled("qkv", ops)
                qkv = ops.linear_t(ops.layer_norm_t(x, self.ln1.weight, self.ln1.bias, self.ln1.eps), wq, m, 3 * W, W, self.qkv.bias,
                                   alpha=aq, out_shape=(B, L, 3 * W))
        else:
            w, a_ = self._w_split("qkv", ops, mode)
            qkv = ops.linear_split(ops.layer_norm_split(x, self.ln1.weight, self.ln1.bias, self.ln1.eps, mode), w, self.qkv.bias, alpha=a_)
        hip_attn = W == 64 * self.heads and L <= ops.ATTENTION_MAX_SEQ
        if hip_attn and rows is None:
            a6 = ops.attention_split(qkv, self.heads, causal, mode)      # the attention kernel stores the split operand itself
        else:
            a = ops.attention(qkv, self.heads, causal) if hip_attn else self._sdpa(qkv, B, L, W, causal)
            if rows is not None:
                a, x = a[rows].contiguous(), x[rows].contiguous()
            a6 = ops.split_operand(a, mode)
        w, a_ = self._w_split("out", ops, mode)
        x = ops.linear_split(a6, w, self.out.bias, residual=x, alpha=a_)
        s, act = self._act_scale(ops)
        if hand in ("block", "fused") and ops.mlp_fused_supported(W, mlp):
            return self._mlp_hand(x, ops, W, mlp)
        w, a_ = self._w_split("fc1", ops, mode)
        h = ops.linear_split(ops.layer_norm_split(x, self.ln2.weight, self.ln2.bias, self.ln2.eps, mode), w,
                             self._fc1_bias_scaled(s), act=act, alpha=s * a_)
        # fc2: one split pass over the [m, mlp] activations (bf16x6: 16 B per element, ~330 us at the headline shape) buys a
        # GEMM of 1 268 us instead of 1 590 us in the tuner -- and 1 820 us inside the step, where the fp32 GEMMs run at lower
        # clocks than in isolation while the 16-bit ones do not: 17.3 k against 16.7 k scores/s on the same box (twice,
        # alternating).  (With only position-independent solutions allowed for this shape the trade was a loss.)
        w, a_ = self._w_split("fc2", ops, mode)
        return ops.linear_split(ops.split_operand(h, mode), w, self.fc2.bias, residual=x, alpha=a_ / s)

    def _w_tiled_ln(self, name, ln, ops, extra):
        """The layer's weight with the LayerNorm `ln` in front of it folded in (ops.fold_layernorm_weight), once per version of
        the four tensors: (packed weight, 1 / wscale, colsum, bias')."""
        lin = getattr(self, name)
        key = (lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version,
               ln.weight.data_ptr(), ln.weight._version, ln.bias.data_ptr(), ln.bias._version, extra)
        cache = self.__dict__.setdefault("_split_cache", {})
        hit = cache.get((name, "tiled_ln"))
        if hit is None or hit[0] != key:
            hit = (key,) + tuple(ops.fold_layernorm_weight(lin.weight, lin.bias, ln.weight, ln.bias, extra))
            cache[(name, "tiled_ln")] = hit
        return hit[1:]

    def forward_chain(self, x, causal, carry=None, emit=True, fp32_out=True):
        """forward(x, causal) for a run of consecutive blocks: -> (x_out, carry_out).  Where the hand-written GEMMs run the
        whole block (LEMON_GEMM=f16x3, LEMON_MLP=block) the two LayerNorms are FOLDED into them (ops.ln_fold_enabled): `carry` =
        (this block's input as the tile-major operand, its row-statistics partials) as the previous block's fc2 left them (None:
        made here with one pass), and with `emit` this block's fc2 leaves the same for the next one."""
        from . import ops
        mlp = self.fc1.weight.shape[0]
        if x is None:            # (full lean form: the block in front left its result as the operand only; it checked `ok`)
            B, L, W = carry[2]
            ok = True
        else:
            B, L, W = x.shape
            ok = (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and W % 32 == 0 and mlp % 32 == 0 and W <= 1024
                  and ops.gemm_mode() == "f16x3" and ops.mlp_mode() == "block" and ops.ln_fold_enabled()
                  and ops.block_fused_supported(W, mlp, self.heads, L))
        if not ok:
            return self(x, causal), None
        ops.select_attention_arithmetic("f16x3")
        m = B * L
        if carry is None:
            xt, aff = ops.rowstats_t(x, self.ln1.eps)
        else:
            xt, aff = carry[0], ops.ln_finalize(carry[1], m, W, self.ln1.eps)
        wq, aq, csq, bq = self._w_tiled_ln("qkv", self.ln1, ops, 1.0)
        qkv = ops.linear_t_ln(xt, wq, m, 3 * W, W, bq, alpha=aq, out_shape=(B, L, 3 * W), row_aff=aff, colsum=csq)
        wo, ao = self._w_tiled("out", ops)
        # inside the chain the residual stream exists as the GEMMs' operand only (ops.chain_operand_residual): the next GEMM reads
        # it as its operand, the one behind it as its residual; fp32 is written where the caller asks for it (fp32_out: the last
        # chained block)
        lean = ops.chain_operand_residual()
        res = dict(residual_t=xt) if lean == 2 else dict(residual=x)
        x, xt, st = ops.linear_t_chain(ops.attention_t(qkv, self.heads, causal), wo, m, W, W, self.out.bias, alpha=ao,
                                       out_shape=(B, L, W), fp32_out=not lean, **res)
        aff = ops.ln_finalize(st, m, W, self.ln2.eps)
        s, act = self._act_scale(ops)
        w1, a1, cs1, b1 = self._w_tiled_ln("fc1", self.ln2, ops, s)
        ht = ops.linear_t_ln(xt, w1, m, mlp, W, b1, act=act, alpha=s * a1, row_aff=aff, colsum=cs1)
        w2, a2 = self._w_tiled("fc2", ops)
        res = dict(residual_t=xt) if lean else dict(residual=x)
        x, xt, st = ops.linear_t_chain(ht, w2, m, W, mlp, self.fc2.bias, alpha=a2 / s, out_shape=(B, L, W), fp32_out=fp32_out or lean < 2, **res)
        return x, ((xt, st, (B, L, W)) if emit else None)

    def _sdpa(self, qkv, B, L, W, causal):
        q, k, v = qkv.view(B, L, 3, self.heads, W // self.heads).permute(2, 0, 3, 1, 4)
        return F.scaled_dot_product_attention(q, k, v, is_causal=causal).transpose(1, 2).reshape(B, L, W)


class VisionTower(nn.Module):
    """CLIP's ViT (HF CLIPVisionTransformer / chexzero_clip.py:226-260).  With act='gelu', patch_bias=True, pre_ln=False it is
    timm's `vit_base_patch16_224` as open_clip wraps it for BiomedCLIP (lemon_amd/biomed.py): a convolution bias, no LayerNorm
    in front of the blocks, exact GELU -- same token layout, same pre-LN blocks, CLS pooling behind the final norm."""

    def __init__(self, cfg, act="quick_gelu", patch_bias=False, pre_ln=True, eps=None):
        super().__init__()
        v = cfg.vision
        eps = cfg.layer_norm_eps if eps is None else eps
        self.patch = nn.Conv2d(3, v.width, cfg.patch_size, cfg.patch_size, bias=patch_bias)
        n_pos = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.cls = nn.Parameter(torch.zeros(v.width))
        self.pos = nn.Parameter(torch.zeros(n_pos, v.width))
        self.pre_ln = nn.LayerNorm(v.width, eps=eps) if pre_ln else None
        self.blocks = nn.ModuleList([Block(v, eps, act) for _ in range(v.layers)])
        self.post_ln = nn.LayerNorm(v.width, eps=eps)
        self.proj = nn.Linear(v.width, cfg.embed_dim, bias=False)

    def _pos_for_gemm(self):
        """The position embedding the token-assembly kernel adds behind a patch-embedding GEMM: with a convolution bias (timm)
        the bias rides on the patch tokens' rows (the GEMM paths run the convolution without it), once per parameter version."""
        if self.patch.bias is None:
            return self.pos
        key = (self.pos.data_ptr(), self.pos._version, self.patch.bias.data_ptr(), self.patch.bias._version)
        cache = self.__dict__.setdefault("_split_cache", {})
        hit = cache.get("pos_bias")
        if hit is None or hit[0] != key:
            pos = self.pos.detach().clone()
            pos[1:] += self.patch.bias.detach()
            hit = (key, pos)
            cache["pos_bias"] = hit
        return hit[1]

    def forward(self, pixel_values):
        conv = False
        if hasattr(pixel_values, "at"):
            # data.PatchOperand: the patch rows are already the tile-major fp16 operand (lemon_preprocess_u8_f16x3t) -- the patch
            # embedding runs in the hand-written GEMM, no fp32 pixel tensor and no split pass exist
            from . import ops
            po = pixel_values
            w = self.patch.weight.reshape(self.patch.weight.shape[0], -1)
            cache = self.__dict__.setdefault("_split_cache", {})
            hit = cache.get(("patch", "tiled"))
            if hit is None or hit[0] != (w.data_ptr(), self.patch.weight._version):
                wscale = ops.weight_scale_f16x3(w)
                hit = ((w.data_ptr(), self.patch.weight._version), ops.pack_weight_t(w.detach().contiguous(), wscale), 1.0 / wscale)
                cache[("patch", "tiled")] = hit
            x = ops.linear_t(po.at, hit[1], po.batch * po.n_patches, w.shape[0], po.k, alpha=hit[2],
                             out_shape=(po.batch, po.n_patches, w.shape[0]))
        elif pixel_values.dim() == 3:
            # patch-major input [B, nP, 3*P*P] (lemon_preprocess_u8 with patch=P): the stride == kernel
            # convolution is a plain GEMM with the flattened filter bank, no im2col and no MIOpen
            from . import ops
            w = self.patch.weight.reshape(self.patch.weight.shape[0], -1)
            mode = ops.gemm_mode()
            if mode != "f32" and pixel_values.is_cuda and not torch.is_grad_enabled() and w.shape[1] % 4 == 0:
                # as a split GEMM too: one split pass over the pixels (normalised pixels are O(1): far inside fp16) buys a
                # GEMM at 2x the fp32 one's rate (k = 3 P^2 = 3 072 for ViT-B/32)
                ws, a_ = split_weight_cached(self, "patch", w, ops, mode)
                x = ops.linear_split(ops.split_operand(pixel_values, mode), ws, alpha=a_)
            else:
                x = ops.linear(pixel_values, w)
        else:
            x = self.patch(pixel_values.to(self.patch.weight.dtype)).flatten(2).transpose(1, 2)
            conv = True                                   # (the convolution added its own bias)
        pos = self.pos if conv else self._pos_for_gemm()
        fused = x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[-1] % 4 == 0
        if fused:     # class token + position embedding (+ pre-LayerNorm) in one pass over the token matrix
            from . import ops
            if self.pre_ln is not None:
                x = ops.vision_tokens_ln(x, self.cls, pos, self.pre_ln.weight, self.pre_ln.bias, self.pre_ln.eps)
            else:
                x = ops.vision_tokens_ln(x, self.cls, pos, None, None)
        else:
            x = torch.cat([self.cls.expand(x.shape[0], 1, -1), x], dim=1) + pos
            if self.pre_ln is not None:
                x = self.pre_ln(x)
        carry = None
        n_chain = len(self.blocks) - 1
        for i, b in enumerate(self.blocks[:-1]):
            x, carry = b.forward_chain(x, causal=False, carry=carry, fp32_out=i == n_chain - 1)
        batch = torch.arange(x.shape[0], device=x.device)
        x = self.blocks[-1](x, causal=False, rows=(batch, torch.zeros_like(batch)), carry=carry)   # CLS rows of the last block
        if fused:
            return ops.linear(ops.layer_norm(x, self.post_ln.weight, self.post_ln.bias, self.post_ln.eps), self.proj.weight)
        return self.proj(self.post_ln(x))


class TextTower(nn.Module):
    def __init__(self, cfg: ClipConfig):
        super().__init__()
        t = cfg.text
        self.eos_token_id = cfg.eos_token_id
        self.tok = nn.Embedding(cfg.vocab_size, t.width)
        self.pos = nn.Parameter(torch.zeros(cfg.context_length, t.width))
        self.blocks = nn.ModuleList([Block(t, cfg.layer_norm_eps) for _ in range(t.layers)])
        self.final_ln = nn.LayerNorm(t.width, eps=cfg.layer_norm_eps)
        self.proj = nn.Linear(t.width, cfg.embed_dim, bias=False)

    def last_token_index(self, input_ids):
        """a caption's EOT position = argmax of its ids (EOT has the largest id: chexzero_clip.py:374-376)"""
        return input_ids.argmax(dim=-1)

    def seq_len_for(self, eot_max):
        """Token count the tower runs for a batch whose last EOT sits at `eot_max`: the longest prompt rounded up
        to a multiple of 8 (capped at the context length).  Truncation is exact under the causal mask; the
        bucket keeps the GEMM row count m = B*L to a handful of values, so caption batches do not meet a new
        (m,n,k) solution key -- and a new summation order -- for every prompt length."""
        return min(self.pos.shape[0], (int(eot_max) + 8) // 8 * 8)

    def forward(self, input_ids, seq_len=None):
        # EOT position = argmax of the ids (EOT has the largest id: chexzero_clip.py:374-376 and HF's
        # legacy eos path); truncate the batch to the longest prompt (exact under the causal mask).
        # seq_len: precomputed on the host by the caller (pipeline.Embedder) to avoid a device sync here.
        eot = input_ids.argmax(dim=-1)
        L = self.seq_len_for(eot.max().item()) if seq_len is None else int(seq_len)
        fused = input_ids.is_cuda and self.pos.dtype == torch.float32 and not torch.is_grad_enabled() and self.pos.shape[-1] % 4 == 0
        if fused:     # embedding lookup + position embedding in one pass, straight from the [B, ctx] id matrix
            from . import ops
            x = ops.text_tokens(input_ids if input_ids.dtype == torch.int64 and input_ids.stride(1) == 1 else input_ids.long().contiguous(),
                                L, self.tok.weight, self.pos)
        else:
            x = self.tok(input_ids[:, :L]) + self.pos[:L]
        carry = None
        n_chain = len(self.blocks) - 1
        for i, b in enumerate(self.blocks[:-1]):
            x, carry = b.forward_chain(x, causal=True, carry=carry, fp32_out=i == n_chain - 1)
        x = self.blocks[-1](x, causal=True, rows=(torch.arange(x.shape[0], device=x.device), eot), carry=carry)   # EOT rows only
        if fused:
            return ops.linear(ops.layer_norm(x, self.final_ln.weight, self.final_ln.bias, self.final_ln.eps), self.proj.weight)
        return self.proj(self.final_ln(x))


class LemonCLIP(nn.Module):
    """encode_text / encode_image with the signatures of HuggingfaceCLIPModel
    (lib/models/downstream_models.py:37-41).  Outputs are un-normalised [B, embed_dim]."""

    def __init__(self, cfg: ClipConfig = None):
        super().__init__()
        self.cfg = cfg or ClipConfig()
        self.vision = VisionTower(self.cfg)
        self.text = TextTower(self.cfg)
        self.logit_scale = math.log(1 / 0.07)          # CLIP's temperature parameter (log scale), chexzero_clip.py:319
        self.reset_parameters()

    def reset_parameters(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        for p in self.parameters():
            if p.dim() > 1:
                with torch.no_grad():
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        for m in self.modules():
            if isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)
        with torch.no_grad():
            self.vision.cls.copy_(torch.randn(self.vision.cls.shape, generator=g) * 0.02)

    @torch.no_grad()
    def encode_image(self, pixel_values=None):
        return self.vision(pixel_values)

    @torch.no_grad()
    def encode_text(self, input_ids=None, attention_mask=None, seq_len=None):
        # attention_mask is accepted for signature parity and ignored: EOT-pooled features of a causal
        # transformer do not depend on it (max |delta| = 0.0 measured)
        return self.text(input_ids, seq_len=seq_len)

    @torch.no_grad()
    def encode_text_dedup(self, input_ids):
        """Embed each distinct prompt once and gather (identical rows give identical embeddings)."""
        uniq, inv = torch.unique(input_ids, dim=0, return_inverse=True)
        return self.text(uniq)[inv]

    # ---------------------------------------------------------------- HF checkpoint mapping
    def load_hf_state_dict(self, sd):
        """Load a `transformers.CLIPModel` state dict (names as in openai/clip-vit-* checkpoints)."""
        own = {}

        def tower(prefix, blocks, n):
            for i in range(n):
                p = f"{prefix}.encoder.layers.{i}."
                b = f"{blocks}.{i}."
                for kind in ("weight", "bias"):
                    own[b + f"qkv.{kind}"] = torch.cat([sd[p + f"self_attn.{x}_proj.{kind}"] for x in "qkv"], 0)
                    own[b + f"out.{kind}"] = sd[p + f"self_attn.out_proj.{kind}"]
                    own[b + f"ln1.{kind}"] = sd[p + f"layer_norm1.{kind}"]
                    own[b + f"ln2.{kind}"] = sd[p + f"layer_norm2.{kind}"]
                    own[b + f"fc1.{kind}"] = sd[p + f"mlp.fc1.{kind}"]
                    own[b + f"fc2.{kind}"] = sd[p + f"mlp.fc2.{kind}"]

        tower("vision_model", "vision.blocks", self.cfg.vision.layers)
        tower("text_model", "text.blocks", self.cfg.text.layers)
        own["vision.patch.weight"] = sd["vision_model.embeddings.patch_embedding.weight"]
        own["vision.cls"] = sd["vision_model.embeddings.class_embedding"]
        own["vision.pos"] = sd["vision_model.embeddings.position_embedding.weight"]
        for kind in ("weight", "bias"):
            own[f"vision.pre_ln.{kind}"] = sd[f"vision_model.pre_layrnorm.{kind}"]
            own[f"vision.post_ln.{kind}"] = sd[f"vision_model.post_layernorm.{kind}"]
            own[f"text.final_ln.{kind}"] = sd[f"text_model.final_layer_norm.{kind}"]
        own["vision.proj.weight"] = sd["visual_projection.weight"]
        own["text.tok.weight"] = sd["text_model.embeddings.token_embedding.weight"]
        own["text.pos"] = sd["text_model.embeddings.position_embedding.weight"]
        own["text.proj.weight"] = sd["text_projection.weight"]
        missing, unexpected = self.load_state_dict(own, strict=True), None
        return self

    # ---------------------------------------------------------------- OpenAI-format checkpoint mapping
    def load_openai_state_dict(self, sd):
        """Load a state dict with the names of OpenAI CLIP / the in-tree copy (lib/models/chexzero_clip.py:263-392):
        what `load_clip(model_path)` (:458-479) and `clip.load(...)` + `load_state_dict` (lib/models/utils.py:94-97)
        consume.  nn.MultiheadAttention's packed in_proj is already this module's fused QKV layout; the two
        projections are stored [width, embed_dim] there (x @ proj) and transposed here."""
        own = {}

        def tower(prefix, blocks, n):
            for i in range(n):
                p, b = f"{prefix}transformer.resblocks.{i}.", f"{blocks}.{i}."
                own[b + "qkv.weight"], own[b + "qkv.bias"] = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
                for kind in ("weight", "bias"):
                    own[b + f"out.{kind}"] = sd[p + f"attn.out_proj.{kind}"]
                    own[b + f"ln1.{kind}"] = sd[p + f"ln_1.{kind}"]
                    own[b + f"ln2.{kind}"] = sd[p + f"ln_2.{kind}"]
                    own[b + f"fc1.{kind}"] = sd[p + f"mlp.c_fc.{kind}"]
                    own[b + f"fc2.{kind}"] = sd[p + f"mlp.c_proj.{kind}"]

        tower("visual.", "vision.blocks", self.cfg.vision.layers)
        tower("", "text.blocks", self.cfg.text.layers)
        own["vision.patch.weight"] = sd["visual.conv1.weight"]
        own["vision.cls"] = sd["visual.class_embedding"]
        own["vision.pos"] = sd["visual.positional_embedding"]
        for kind in ("weight", "bias"):
            own[f"vision.pre_ln.{kind}"] = sd[f"visual.ln_pre.{kind}"]
            own[f"vision.post_ln.{kind}"] = sd[f"visual.ln_post.{kind}"]
            own[f"text.final_ln.{kind}"] = sd[f"ln_final.{kind}"]
        own["vision.proj.weight"] = sd["visual.proj"].t()
        own["text.tok.weight"] = sd["token_embedding.weight"]
        own["text.pos"] = sd["positional_embedding"]
        own["text.proj.weight"] = sd["text_projection"].t()
        self.load_state_dict({k: v.float() for k, v in own.items()}, strict=True)
        if "logit_scale" in sd:
            self.logit_scale = float(sd["logit_scale"])
        return self

    @classmethod
    def from_openai_checkpoint(cls, path, cfg=None):
        """A LOCAL `.pt` holding an OpenAI-format state dict (the files lib/models/utils.py:20-25 points at)."""
        if not os.path.isfile(path):
            raise FileNotFoundError(
                f"CLIP checkpoint {path!r} not found. The reference loads its in-tree CLIP variants from the authors' cluster "
                "paths (lib/models/utils.py:20-25); pass a local OpenAI-format state dict with --clip_path.")
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
        sd = {k: v for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
        return cls(cfg or ClipConfig.from_openai_state_dict(sd)).load_openai_state_dict(sd)

    @property
    def context_length(self):
        return self.cfg.context_length

    @classmethod
    def from_pretrained(cls, path):
        """Local HF checkpoint directory only (no hub access in this environment)."""
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"CLIP weights {path!r}: not a local directory. The reference loads "
                "'openai/clip-vit-base-patch32' from the HF hub (lib/models/utils.py:66-67); this build has no "
                "network, so pass a local checkpoint directory (--clip_path) or use random weights (--clip_path random).")
        with open(os.path.join(path, "config.json")) as f:
            cfg = ClipConfig.from_hf_dict(json.load(f))
        model = cls(cfg)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        return model.load_hf_state_dict(sd)


class SyntheticTokenizer:
    """Deterministic stand-in for the HF CLIP tokenizer when no vocabulary files exist locally
    (synthetic runs, smoke tests): whitespace words -> stable ids, SOT/EOT framing, max_length padding.
    Callable like the reference uses it: tokenizer(list[str], padding="max_length", truncation=True)
    -> {"input_ids": [[...]], "attention_mask": [[...]]}  (run_lemon.py:151-154)."""

    def __init__(self, vocab_size=49408, context_length=77, eos_token_id=49407, sot_token_id=None, word_ids=None):
        self.vocab_size, self.context_length, self.eos = vocab_size, context_length, eos_token_id
        self.sot = eos_token_id - 1 if sot_token_id is None else sot_token_id
        # word ids fall into [lo, hi): below the SOT / EOT pair by default (CLIP: they are the two largest ids)
        self.word_ids = (1, self.sot) if word_ids is None else word_ids

    def _word_id(self, w):
        h = 2166136261
        for ch in w.lower().encode("utf-8"):
            h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
        lo, hi = self.word_ids
        return lo + h % (hi - lo)

    def __call__(self, texts, padding="max_length", truncation=True, **_):
        ids, mask = [], []
        for t in texts:
            toks = [self.sot] + [self._word_id(w) for w in str(t).split()][: self.context_length - 2] + [self.eos]
            pad = self.context_length - len(toks)
            ids.append(toks + [0] * pad)
            mask.append([1] * len(toks) + [0] * pad)
        return {"input_ids": ids, "attention_mask": mask}


IN_TREE_CLIP = {   # lib/models/utils.py:82-103: branch -> (architecture of the checkpoint it loads, context length)
    "mimic_clip_from_scratch_random": "chexzero-scratch-256", "mimic_clip_from_scratch_cat": "chexzero-scratch-256",
    "cc3m_clip_from_scratch": "chexzero-scratch-77", "chexzero": "vit-b-32",
}


def _bpe_tokenizer(bpe_path):
    from .tokenizer import ClipBPE, find_bpe_file
    f = find_bpe_file(bpe_path)
    return ClipBPE.from_file(f) if f else None


def algorithm_class_from_scratch(name, text_base_name="openai/clip-vit-base-patch32", img_base=None,
                                 return_tokenizer=False, arch=None, bpe_path=None):
    """lib/models/utils.py:64-105.  Returns `model` or `(model, tokenizer)` with the reference's two calling
    conventions:
      'huggingface_clip'             model.encode_text(input_ids, attention_mask); tokenizer(texts, padding=
                                     "max_length", truncation=True) -> dict of lists          (:65-71)
      'mimic_clip_from_scratch_*', 'cc3m_clip_from_scratch', 'chexzero'
                                     model.encode_text(tokens); tokenizer(texts) -> LongTensor [n, ctx]  (:82-103,
                                     lib/models/chexzero_clip.py:458-493)
    `text_base_name` is LOCAL: an HF checkpoint directory (huggingface_clip), an OpenAI-format `.pt` state dict (the
    in-tree branches; the reference hard-codes the authors' cluster paths, :20-25), or 'random[:arch]' for seeded
    random weights.  Tokenizer: the HF tokenizer files of the checkpoint directory when present, else a CLIP BPE
    built from a user-supplied merges file (`bpe_path` / $LEMON_BPE_PATH), else -- random weights only -- the
    hash-based SyntheticTokenizer.
      'biomed_clip'                  open_clip's BiomedCLIP (:72-78): same calling convention as the in-tree branches;
                                     `text_base_name` = a local copy of the hub snapshot (open_clip_pytorch_model.bin + vocab.txt)
                                     or 'random[:biomed-tiny]'; WordPiece vocabulary from there, `bpe_path` or $LEMON_VOCAB_PATH."""
    rand = str(text_base_name).startswith("random")
    parts = str(text_base_name).split(":")
    if name == "huggingface_clip":
        if rand:
            model = LemonCLIP(ClipConfig.named(arch or (parts[1] if len(parts) > 1 else "vit-b-32")))
            bpe = _bpe_tokenizer(bpe_path) if model.cfg.vocab_size == 49408 else None
        else:
            model = LemonCLIP.from_pretrained(text_base_name)
            bpe = None
        if not return_tokenizer:
            return model
        cfg = model.cfg
        if not rand and any(os.path.exists(os.path.join(text_base_name, f)) for f in ("tokenizer.json", "vocab.json")):
            from transformers import AutoTokenizer
            tok = AutoTokenizer.from_pretrained(text_base_name, local_files_only=True)
        else:
            bpe = bpe or (_bpe_tokenizer(bpe_path or (None if rand else text_base_name)) if cfg.vocab_size == 49408 else None)
            if bpe is not None:
                from .tokenizer import HFStyleClipTokenizer
                tok = HFStyleClipTokenizer(bpe, cfg.context_length)
            elif rand:
                tok = SyntheticTokenizer(cfg.vocab_size, cfg.context_length, cfg.eos_token_id)
            else:
                raise FileNotFoundError(f"no tokenizer files in {text_base_name!r} and no merges file (--bpe_path / LEMON_BPE_PATH)")
        return model, tok
    if name in IN_TREE_CLIP:
        if rand:
            model = LemonCLIP(ClipConfig.named(arch or (parts[1] if len(parts) > 1 else IN_TREE_CLIP[name])))
        else:
            # mimic/cc3m: load_clip(model_path, context_length) builds the fixed architecture then loads the weights
            # (:458-479); chexzero: clip.load("ViT-B/32") + load_state_dict (:94-97) -- both = the checkpoint's shapes
            model = LemonCLIP.from_openai_checkpoint(text_base_name)
            want = ClipConfig.named(IN_TREE_CLIP[name])
            if (model.cfg.embed_dim, model.cfg.patch_size, model.cfg.context_length) != (want.embed_dim, want.patch_size, want.context_length):
                raise ValueError(f"{text_base_name!r} is not a {name} checkpoint: embed_dim/patch/context "
                                 f"{(model.cfg.embed_dim, model.cfg.patch_size, model.cfg.context_length)} != "
                                 f"{(want.embed_dim, want.patch_size, want.context_length)}")
        if not return_tokenizer:
            return model
        bpe = _bpe_tokenizer(bpe_path) if model.cfg.vocab_size == 49408 else None
        if bpe is not None:
            from .tokenizer import tokenize
            tok = lambda texts: tokenize(texts, model, bpe)
        elif rand:
            syn = SyntheticTokenizer(model.cfg.vocab_size, model.cfg.context_length, model.cfg.eos_token_id)
            tok = lambda texts: torch.tensor(syn(texts)["input_ids"], dtype=torch.long)
        else:
            raise FileNotFoundError("the in-tree CLIP branches tokenise with the CLIP BPE (chexzero_clip.py:481-493): "
                                    "supply its merges file with --bpe_path or LEMON_BPE_PATH")
        return model, tok
    if name == "biomed_clip":
        # lib/models/utils.py:72-78: open_clip's BiomedCLIP (timm ViT-B/16 + PubMedBERT, lemon_amd/biomed.py); tokenizer(texts)
        # -> LongTensor [n, 256], model.encode_text(tokens) (run_lemon.py:148-160)
        from .biomed import BiomedCLIP, BiomedConfig
        from .tokenizer import BertWordPiece, find_vocab_file
        if rand:
            model = BiomedCLIP(BiomedConfig.named(arch or (parts[1] if len(parts) > 1 else "biomed")))
        else:
            model = BiomedCLIP.from_pretrained(text_base_name)
        if not return_tokenizer:
            return model
        vf = find_vocab_file(bpe_path) or (None if rand else find_vocab_file(text_base_name))
        if vf is not None:
            tok = BertWordPiece.from_file(vf, model.cfg.context_length)
            if len(tok.vocab) > model.cfg.vocab_size:
                raise ValueError(f"{vf!r} holds {len(tok.vocab)} tokens, the model's embedding {model.cfg.vocab_size}")
        elif rand:
            # hash-based stand-in with BERT's framing: [CLS] = 2, [SEP] = 3, [PAD] = 0 (synthetic runs only)
            syn = SyntheticTokenizer(model.cfg.vocab_size, model.cfg.context_length, eos_token_id=3, sot_token_id=2,
                                     word_ids=(4, model.cfg.vocab_size))
            tok = lambda texts: torch.tensor(syn(texts)["input_ids"], dtype=torch.long)
        else:
            raise FileNotFoundError(f"no vocab.txt beside {text_base_name!r}: supply the BERT vocabulary with --bpe_path or LEMON_VOCAB_PATH")
        return model, tok
    raise NotImplementedError(name)


def encoder_flops(cfg: ClipConfig, n_tokens_text=None):
    """Forward FLOPs per (image, caption) pair as executed: 2*m*n*k per GEMM + attention; the last block's
    output projection and MLP run for the pooled token only (Block.forward(rows=...))."""
    def tower(t: TowerConfig, L):
        attn = 2 * L * 3 * t.width * t.width + 4 * L * L * t.width            # QKV projection + attention
        rowwise = 2 * (t.width * t.width + 2 * t.width * t.mlp)                # out-proj + MLP, per token
        return t.layers * attn + ((t.layers - 1) * L + 1) * rowwise
    Lv = (cfg.image_size // cfg.patch_size) ** 2 + 1
    Lt = n_tokens_text or cfg.context_length
    img = tower(cfg.vision, Lv) + 2 * (Lv - 1) * 3 * cfg.patch_size ** 2 * cfg.vision.width + 2 * cfg.vision.width * cfg.embed_dim
    txt = tower(cfg.text, Lt) + 2 * cfg.text.width * cfg.embed_dim
    return img, txt
