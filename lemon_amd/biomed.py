"""BiomedCLIP -- the `--clip_model biomed_clip` branch of the model factory (lib/models/utils.py:72-78; the reference's
mimic-cxr experiments run on it: experiments.py:73,123,161,200, lib/datasets/utils.py:396).

The reference gets the model from a third-party package that is NOT under /root/reference and not installed here:
    open_clip (open_clip_torch, un-pinned in requirements.txt) `create_model_from_pretrained('hf-hub:microsoft/
    BiomedCLIP-PubMedBERT_256-vit_base_patch16_224')` + `get_tokenizer(...)`
so what is restated below is that package's published architecture for this checkpoint [recollection, not executable here]:
    vision   timm `vit_base_patch16_224` (open_clip TimmModel, pool '' -> the model's own CLS-token pooling, proj 'linear'):
             patch convolution WITH bias, [cls; patches] + pos_embed, NO LayerNorm in front of the blocks, 12 pre-LN blocks
             (eps 1e-6, exact GELU), final norm, CLS row, bias-free Linear(768, 512)
    text     HF `BertModel` (microsoft/BiomedNLP-PubMedBERT-base-uncased-abstract, no pooler) inside open_clip's HFTextEncoder:
             word + position + token-type embeddings -> LayerNorm(1e-12) -> 12 POST-LN layers (bidirectional attention with the
             padding mask `ids != pad`, exact GELU) -> hidden state of [CLS] (`cls_last_hidden_state_pooler`) -> proj 'mlp':
             Linear(768, 640, no bias) -> GELU -> Linear(640, 512, no bias)
    tokens   BERT uncased WordPiece, [CLS] ... [SEP], padded to context_length 256 (lemon_amd/tokenizer.BertWordPiece)
Parity is anchored where it can be: the vision tower against HF `ViTModel` and the text tower against HF `BertModel` (both
importable here, seeded random weights, tests/test_biomed.py), the call sites against run_lemon.py:148-160 (tokenizer(texts) ->
LongTensor, model.encode_text(tokens), model.encode_image(pixels)).  Real-weight parity needs a local checkpoint directory.

MI355X-first: the vision tower IS lemon_amd.clip.VisionTower (same hand-written GEMM chain, with the GELU operand epilogue and
the convolution bias riding on the position embedding).  The text tower's padding mask never reaches a kernel: captions are
grouped by their exact token count and every group runs un-padded (for bidirectional attention a key mask over trailing pads
and truncation to the caption's own length are the same function) -- pipeline.Embedder sorts captions by length so that a
micro-batch is one group.  The post-LN layers reuse the LayerNorm fold of the hand-written GEMMs: the output projection / fc2
write the pre-norm sum y as fp32, as the next GEMM's operand and as row statistics (EMIT); fc1 / the next layer's QKV take y with
the LayerNorm folded into their weights (FOLD); the LayerNorm kernel only makes the fp32 residual.
"""
import json
import os
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

from .clip import Block, TowerConfig, VisionTower, split_weight_cached


@dataclass
class BiomedConfig:
    embed_dim: int = 512
    image_size: int = 224
    patch_size: int = 16
    vision: TowerConfig = field(default_factory=lambda: TowerConfig(768, 12, 12, 3072))
    text: TowerConfig = field(default_factory=lambda: TowerConfig(768, 12, 12, 3072))
    vocab_size: int = 30522
    context_length: int = 256           # open_clip text_cfg.context_length: the tokenizer's max_length
    max_positions: int = 512            # BERT max_position_embeddings
    type_vocab_size: int = 2
    pad_token_id: int = 0
    proj_hidden: int = 640              # open_clip 'mlp' projection: (width + embed_dim) // 2
    layer_norm_eps: float = 1e-6        # timm ViT
    text_layer_norm_eps: float = 1e-12  # BERT

    @staticmethod
    def named(name):
        name = str(name).lower().replace("_", "-")
        if name in ("biomed", "biomedclip", "biomed-clip", "pubmedbert-256-vit-b-16"):
            return BiomedConfig()
        if name in ("biomed-tiny", "tiny", "test"):
            return BiomedConfig(embed_dim=32, image_size=32, patch_size=8, vision=TowerConfig(64, 2, 1, 128),
                                text=TowerConfig(64, 2, 1, 128), vocab_size=300, context_length=24, max_positions=32,
                                proj_hidden=48)
        raise ValueError(f"unknown BiomedCLIP architecture {name!r}")

    @staticmethod
    def from_open_clip_state_dict(sd, context_length=256):
        """The architecture read off the tensor shapes of an open_clip-format state dict (heads = width / 64, as in both towers
        of the published checkpoint)."""
        pw = sd["visual.trunk.patch_embed.proj.weight"]
        vw, patch = pw.shape[0], pw.shape[-1]
        grid = round((sd["visual.trunk.pos_embed"].shape[-2] - 1) ** 0.5)
        vl = len([k for k in sd if k.startswith("visual.trunk.blocks.") and k.endswith(".attn.qkv.weight")])
        tw = sd["text.transformer.embeddings.word_embeddings.weight"].shape[1]
        tl = len([k for k in sd if k.startswith("text.transformer.encoder.layer.") and k.endswith(".attention.self.query.weight")])
        return BiomedConfig(embed_dim=sd["visual.head.proj.weight"].shape[0], image_size=patch * grid, patch_size=patch,
                            vision=TowerConfig(vw, vl, max(1, vw // 64), sd["visual.trunk.blocks.0.mlp.fc1.weight"].shape[0]),
                            text=TowerConfig(tw, tl, max(1, tw // 64), sd["text.transformer.encoder.layer.0.intermediate.dense.weight"].shape[0]),
                            vocab_size=sd["text.transformer.embeddings.word_embeddings.weight"].shape[0],
                            context_length=min(context_length, sd["text.transformer.embeddings.position_embeddings.weight"].shape[0]),
                            max_positions=sd["text.transformer.embeddings.position_embeddings.weight"].shape[0],
                            type_vocab_size=sd["text.transformer.embeddings.token_type_embeddings.weight"].shape[0],
                            proj_hidden=sd["text.proj.0.weight"].shape[0])


class BertLayer(Block):
    """One HF BertLayer (attention.self + attention.output + intermediate + output): x1 = LN1(x + out(attn(qkv(x)))),
    x2 = LN2(x1 + fc2(gelu(fc1(x1)))).  Parameter names follow clip.Block (ln1 = attention.output.LayerNorm, ln2 =
    output.LayerNorm); no mask: the caller hands over captions without padding."""

    def __init__(self, cfg: TowerConfig, eps):
        super().__init__(cfg, eps, act="gelu")

    def _gemm(self, t, name, ops, mode, residual=None, act=None):
        lin = getattr(self, name)
        if mode == "f32" or t.shape[-1] % 4:
            return ops.linear(t, lin.weight, lin.bias, residual=residual, act=act)
        w, a_ = split_weight_cached(self, name, lin.weight, ops, mode)
        return ops.linear_split(ops.split_operand(t, mode), w, lin.bias, residual=residual, act=act, alpha=a_)

    def forward(self, x, causal=False, rows=None, carry=None):
        B, L, W = x.shape
        if x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled():
            # every GEMM through the library entry points of the current mode (lemon_linear_f32 / _bf16x6 / _f16x3), LayerNorm and
            # attention kernels: the form the fallbacks and the A/B modes run
            from . import ops
            mode = ops.gemm_mode()
            ops.select_attention_arithmetic(mode)
            qkv = self._gemm(x, "qkv", ops, mode)
            if W == 64 * self.heads and L <= ops.ATTENTION_MAX_SEQ:
                a = ops.attention(qkv, self.heads, False)
            else:
                a = self._sdpa(qkv, B, L, W, False)
            if rows is not None:
                a, x = a[rows].contiguous(), x[rows].contiguous()
            x = ops.layer_norm(self._gemm(a, "out", ops, mode, residual=x), self.ln1.weight, self.ln1.bias, self.ln1.eps)
            h = self._gemm(x, "fc1", ops, mode, act="gelu")
            return ops.layer_norm(self._gemm(h, "fc2", ops, mode, residual=x), self.ln2.weight, self.ln2.bias, self.ln2.eps)
        a = self._sdpa(self.qkv(x), B, L, W, False)
        if rows is not None:
            a, x = a[rows], x[rows]
        x = self.ln1(x + self.out(a))
        return self.ln2(x + self.fc2(F.gelu(self.fc1(x))))

    def chain_supported(self, x):
        from . import ops
        B, L, W = x.shape
        mlp = self.fc1.weight.shape[0]
        return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and W % 32 == 0 and mlp % 32 == 0 and W <= 1024
                and ops.gemm_mode() == "f16x3" and ops.mlp_mode() == "block" and ops.ln_fold_enabled()
                and ops.block_fused_supported(W, mlp, self.heads, L))

    def forward_chain(self, x, carry, last=False):
        """x = LN_prev(y_prev) in fp32; carry = (y_prev as the tile-major operand, its rows' (rstd, -mean rstd), LN_prev): QKV
        folds LN_prev.  -> (x_out, carry_out); `last`: -> ([B, W] rows of the [CLS] token, None)."""
        from . import ops
        B, L, W = x.shape
        m, mlp = B * L, self.fc1.weight.shape[0]
        ops.select_attention_arithmetic("f16x3")
        yt, aff, ln_prev = carry
        wq, aq, csq, bq = self._w_tiled_ln("qkv", ln_prev, ops, 1.0)
        qkv = ops.linear_t_ln(yt, wq, m, 3 * W, W, bq, alpha=aq, out_shape=(B, L, 3 * W), row_aff=aff, colsum=csq)
        if last:
            # the tower reads the [CLS] row only: attention still sees every token's keys and values, the output projection and
            # the MLP (3/4 of the layer's GEMM work, all row-wise) run for that row alone -- in the library form, B rows
            a = ops.attention(qkv, self.heads, False)[:, 0].contiguous()
            x = x[:, 0].contiguous()
            x = ops.layer_norm(self._gemm(a, "out", ops, "f16x3", residual=x), self.ln1.weight, self.ln1.bias, self.ln1.eps)
            h = self._gemm(x, "fc1", ops, "f16x3", act="gelu")
            return ops.layer_norm(self._gemm(h, "fc2", ops, "f16x3", residual=x), self.ln2.weight, self.ln2.bias, self.ln2.eps), None
        wo, ao = self._w_tiled("out", ops)
        y, yt, st = ops.linear_t_ln(ops.attention_t(qkv, self.heads, False), wo, m, W, W, self.out.bias, residual=x, alpha=ao,
                                    out_shape=x.shape, emit=True)
        x = ops.layer_norm(y, self.ln1.weight, self.ln1.bias, self.ln1.eps)          # (the residual of the MLP half)
        w1, a1, cs1, b1 = self._w_tiled_ln("fc1", self.ln1, ops, 1.0)
        ht = ops.linear_t_ln(yt, w1, m, mlp, W, b1, act="gelu", alpha=a1, row_aff=ops.ln_finalize(st, m, W, self.ln1.eps), colsum=cs1)
        w2, a2 = self._w_tiled("fc2", ops)
        y, yt, st = ops.linear_t_ln(ht, w2, m, W, mlp, self.fc2.bias, residual=x, alpha=a2, out_shape=x.shape, emit=True)
        return (ops.layer_norm(y, self.ln2.weight, self.ln2.bias, self.ln2.eps),
                (yt, ops.ln_finalize(st, m, W, self.ln2.eps), self.ln2))


class BertTextTower(nn.Module):
    exact_lengths = True      # pipeline.Embedder: micro-batches must not mix caption lengths (no padding mask in the kernels)

    def __init__(self, cfg: BiomedConfig):
        super().__init__()
        t = cfg.text
        self.pad_token_id = cfg.pad_token_id
        self.tok = nn.Embedding(cfg.vocab_size, t.width)
        self.pos = nn.Parameter(torch.zeros(cfg.max_positions, t.width))
        self.type_emb = nn.Parameter(torch.zeros(cfg.type_vocab_size, t.width))
        self.emb_ln = nn.LayerNorm(t.width, eps=cfg.text_layer_norm_eps)
        self.blocks = nn.ModuleList([BertLayer(t, cfg.text_layer_norm_eps) for _ in range(t.layers)])
        self.proj1 = nn.Linear(t.width, cfg.proj_hidden, bias=False)
        self.proj2 = nn.Linear(cfg.proj_hidden, cfg.embed_dim, bias=False)

    # -- the two hooks pipeline.Embedder buckets captions with --
    def last_token_index(self, input_ids):
        """index of a caption's last non-pad token ([SEP]); -1 for an all-pad row"""
        nz = input_ids != self.pad_token_id
        return (nz * torch.arange(1, input_ids.shape[1] + 1, device=input_ids.device)).amax(dim=-1) - 1

    def seq_len_for(self, last):
        return max(1, min(self.pos.shape[0], int(last) + 1))

    def _pos_type0(self):
        """position + token-type-0 embedding (open_clip passes no token_type_ids: all zero), once per parameter version"""
        key = (self.pos.data_ptr(), self.pos._version, self.type_emb.data_ptr(), self.type_emb._version)
        cache = self.__dict__.setdefault("_split_cache", {})
        hit = cache.get("pos_type0")
        if hit is None or hit[0] != key:
            hit = (key, (self.pos.detach() + self.type_emb.detach()[0]).contiguous())
            cache["pos_type0"] = hit
        return hit[1]

    def _masked(self, ids):
        """HF's own formulation (key mask = ids != pad), plain PyTorch: the CPU / autograd path, and on the GPU only rows with
        padding INSIDE the caption (no tokenizer produces them)."""
        B, L = ids.shape
        x = self.emb_ln(self.tok(ids) + self.type_emb[0] + self.pos[:L])
        key_ok = (ids != self.pad_token_id)[:, None, None, :]
        for b in self.blocks:
            W, H = x.shape[-1], b.heads
            q, k, v = b.qkv(x).view(B, L, 3, H, W // H).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(q, k, v, attn_mask=key_ok).transpose(1, 2).reshape(B, L, W)
            x = b.ln1(x + b.out(a))
            x = b.ln2(x + b.fc2(F.gelu(b.fc1(x))))
        return self.proj2(F.gelu(self.proj1(x[:, 0])))

    def _unpadded(self, ids, L):
        """ids [B, >= L], every row exactly L tokens long: the fused GPU path"""
        from . import ops
        ids = ids if ids.dtype == torch.int64 and ids.stride(1) == 1 else ids.long().contiguous()
        e = ops.text_tokens(ids, L, self.tok.weight, self._pos_type0())
        ln = self.emb_ln
        x = ops.layer_norm(e, ln.weight, ln.bias, ln.eps)
        n = len(self.blocks)
        if self.blocks[0].chain_supported(x):
            carry = ops.rowstats_t(e, ln.eps) + (ln,)
            for i, b in enumerate(self.blocks):
                x, carry = b.forward_chain(x, carry, last=i == n - 1)
        else:
            for b in self.blocks[:-1]:
                x = b(x)
            batch = torch.arange(x.shape[0], device=x.device)
            x = self.blocks[-1](x, rows=(batch, torch.zeros_like(batch)))       # [CLS] rows of the last layer
        return ops.linear(ops.linear(x, self.proj1.weight, act="gelu"), self.proj2.weight)

    def forward(self, input_ids, seq_len=None, lengths=None):
        """input_ids [B, ctx] -> [B, embed_dim].  lengths: per-row token counts on the HOST when the caller has them
        (pipeline.Embedder), saving the device read here; seq_len: accepted for the CLIP towers' signature (their bucketed length)."""
        fused = input_ids.is_cuda and self.pos.dtype == torch.float32 and not torch.is_grad_enabled() and self.pos.shape[-1] % 4 == 0
        if not fused:
            L = int(self.last_token_index(input_ids).max()) + 1 if input_ids.shape[0] else 1
            return self._masked(input_ids[:, :max(L, 1)])
        nz = input_ids != self.pad_token_id
        if lengths is None:
            last = self.last_token_index(input_ids)
            inner = (nz.sum(-1) != last + 1)                   # padding inside the caption
            lengths, inner = (last + 1).cpu(), inner.cpu()
        else:
            lengths, inner = torch.as_tensor(lengths).cpu(), torch.zeros(input_ids.shape[0], dtype=torch.bool)
        out = torch.empty((input_ids.shape[0], self.proj2.weight.shape[0]), dtype=torch.float32, device=input_ids.device)
        lengths = lengths.clamp(min=1)
        for L in torch.unique(lengths[~inner]).tolist():
            sel = ((lengths == L) & ~inner).nonzero().flatten()
            rows = input_ids if sel.numel() == input_ids.shape[0] else input_ids[sel.to(input_ids.device)]
            e = self._unpadded(rows, int(L))
            if sel.numel() == input_ids.shape[0]:
                return e
            out[sel.to(out.device)] = e
        if bool(inner.any()):
            sel = inner.nonzero().flatten().to(input_ids.device)
            out[sel] = self._masked(input_ids[sel][:, :int(lengths[inner].max())])
        return out


class BiomedCLIP(nn.Module):
    """encode_image(pixel_values) / encode_text(tokens) as open_clip's CustomTextCLIP is called by run_lemon.py:148-160 with
    normalize=False: un-normalised [B, 512]."""

    def __init__(self, cfg: BiomedConfig = None):
        super().__init__()
        self.cfg = cfg or BiomedConfig()
        self.vision = VisionTower(self.cfg, act="gelu", patch_bias=True, pre_ln=False, eps=self.cfg.layer_norm_eps)
        self.text = BertTextTower(self.cfg)
        self.logit_scale = 0.0
        self.reset_parameters()

    def reset_parameters(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            for m in self.modules():
                if isinstance(m, nn.LayerNorm):
                    nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
                elif isinstance(m, (nn.Linear, nn.Conv2d)) and m.bias is not None:
                    nn.init.zeros_(m.bias)
            self.vision.cls.copy_(torch.randn(self.vision.cls.shape, generator=g) * 0.02)

    @property
    def context_length(self):
        return self.cfg.context_length

    @torch.no_grad()
    def encode_image(self, pixel_values=None):
        return self.vision(pixel_values)

    @torch.no_grad()
    def encode_text(self, input_ids=None, attention_mask=None, seq_len=None, lengths=None):
        # attention_mask: open_clip derives it from the ids (`x != pad_token_id`), so does the tower
        return self.text(input_ids, seq_len=seq_len, lengths=lengths)

    @torch.no_grad()
    def encode_text_dedup(self, input_ids):
        uniq, inv = torch.unique(input_ids, dim=0, return_inverse=True)
        return self.text(uniq)[inv]

    # ---------------------------------------------------------------- open_clip checkpoint names
    def _name_map(self):
        """[(own name, open_clip name, how)] -- how: None (same tensor), 'qkv' (own = cat of three), 'cls' / 'pos' (leading 1-dims)"""
        m = [("vision.patch.weight", "visual.trunk.patch_embed.proj.weight", None), ("vision.patch.bias", "visual.trunk.patch_embed.proj.bias", None),
             ("vision.cls", "visual.trunk.cls_token", "lead"), ("vision.pos", "visual.trunk.pos_embed", "lead"),
             ("vision.post_ln.weight", "visual.trunk.norm.weight", None), ("vision.post_ln.bias", "visual.trunk.norm.bias", None),
             ("vision.proj.weight", "visual.head.proj.weight", None)]
        for i in range(self.cfg.vision.layers):
            o, t = f"vision.blocks.{i}.", f"visual.trunk.blocks.{i}."
            for kind in ("weight", "bias"):
                m += [(o + f"ln1.{kind}", t + f"norm1.{kind}", None), (o + f"qkv.{kind}", t + f"attn.qkv.{kind}", None),
                      (o + f"out.{kind}", t + f"attn.proj.{kind}", None), (o + f"ln2.{kind}", t + f"norm2.{kind}", None),
                      (o + f"fc1.{kind}", t + f"mlp.fc1.{kind}", None), (o + f"fc2.{kind}", t + f"mlp.fc2.{kind}", None)]
        e = "text.transformer.embeddings."
        m += [("text.tok.weight", e + "word_embeddings.weight", None), ("text.pos", e + "position_embeddings.weight", None),
              ("text.type_emb", e + "token_type_embeddings.weight", None), ("text.emb_ln.weight", e + "LayerNorm.weight", None),
              ("text.emb_ln.bias", e + "LayerNorm.bias", None), ("text.proj1.weight", "text.proj.0.weight", None),
              ("text.proj2.weight", "text.proj.2.weight", None)]
        for i in range(self.cfg.text.layers):
            o, t = f"text.blocks.{i}.", f"text.transformer.encoder.layer.{i}."
            for kind in ("weight", "bias"):
                m += [(o + f"qkv.{kind}", tuple(t + f"attention.self.{x}.{kind}" for x in ("query", "key", "value")), "qkv"),
                      (o + f"out.{kind}", t + f"attention.output.dense.{kind}", None), (o + f"ln1.{kind}", t + f"attention.output.LayerNorm.{kind}", None),
                      (o + f"fc1.{kind}", t + f"intermediate.dense.{kind}", None), (o + f"fc2.{kind}", t + f"output.dense.{kind}", None),
                      (o + f"ln2.{kind}", t + f"output.LayerNorm.{kind}", None)]
        return m

    def load_open_clip_state_dict(self, sd):
        """A state dict with open_clip's names for this model (`open_clip_pytorch_model.bin` of the hub snapshot the reference's
        create_model_from_pretrained downloads)."""
        own = {}
        for name, src, how in self._name_map():
            if how == "qkv":
                own[name] = torch.cat([sd[s] for s in src], 0)
            elif how == "lead":
                own[name] = sd[src].reshape(sd[src].shape[-2:] if name.endswith("pos") else sd[src].shape[-1:])
            else:
                own[name] = sd[src]
        self.load_state_dict({k: v.float() for k, v in own.items()}, strict=True)
        if "logit_scale" in sd:
            self.logit_scale = float(sd["logit_scale"])
        return self

    def open_clip_state_dict(self):
        """The inverse mapping (tests; exporting a randomly initialised stand-in)."""
        own, sd = self.state_dict(), {"logit_scale": torch.tensor(float(self.logit_scale))}
        W = self.cfg.text.width
        for name, src, how in self._name_map():
            if how == "qkv":
                for j, s in enumerate(src):
                    sd[s] = own[name][j * W:(j + 1) * W].clone()
            elif how == "lead":
                sd[src] = own[name].reshape((1, 1) + tuple(own[name].shape) if name.endswith("cls") else (1,) + tuple(own[name].shape)).clone()
            else:
                sd[src] = own[name].clone()
        return sd

    @classmethod
    def from_pretrained(cls, path):
        """LOCAL files only: a directory holding open_clip_pytorch_model.bin (| .safetensors) [+ open_clip_config.json], or the
        weight file itself.  The reference downloads them from the HF hub (lib/models/utils.py:73); there is no network here."""
        f = path
        if os.path.isdir(path):
            cands = [os.path.join(path, n) for n in ("open_clip_pytorch_model.bin", "open_clip_model.safetensors", "open_clip_pytorch_model.safetensors")]
            f = next((c for c in cands if os.path.exists(c)), None)
        if f is None or not os.path.isfile(f):
            raise FileNotFoundError(
                f"BiomedCLIP weights {path!r}: no open_clip_pytorch_model.bin there.  The reference loads 'hf-hub:microsoft/"
                "BiomedCLIP-PubMedBERT_256-vit_base_patch16_224' from the hub (lib/models/utils.py:73); this build has no network, so pass "
                "a local copy of that snapshot with --clip_path (or 'random' for seeded random weights).")
        ctx = 256
        cj = os.path.join(os.path.dirname(f), "open_clip_config.json")
        if os.path.exists(cj):                        # the tokenizer's max_length is the one thing the tensors do not tell
            with open(cj) as fh:
                ctx = json.load(fh).get("model_cfg", {}).get("text_cfg", {}).get("context_length", ctx)
        if f.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(f)
        else:
            sd = torch.load(f, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        return cls(BiomedConfig.from_open_clip_state_dict(sd, ctx)).load_open_clip_state_dict(sd)
