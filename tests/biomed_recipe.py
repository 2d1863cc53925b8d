"""Shared by tests/test_biomed.py (CPU) and tests/test_gpu_biomed.py: an independent restatement of BiomedCLIP's two towers from
HF transformers' own modules -- `ViTModel` (= timm's vit_base_patch16_224 math: conv patch embedding with bias, class token,
learned positions, pre-LN blocks with exact GELU, final LayerNorm, CLS row) and `BertModel` (the class open_clip's HFTextEncoder
instantiates for PubMedBERT) -- on seeded random weights, and the mapping of their parameters onto lemon_amd.biomed.BiomedCLIP
through open_clip's checkpoint names (so the loader is exercised too)."""
import torch

from lemon_amd.biomed import BiomedCLIP, BiomedConfig
from lemon_amd.clip import TowerConfig


def config(size):
    if size == "tiny":
        return BiomedConfig.named("biomed-tiny")
    if size == "mid":        # every width a multiple of 256: the hand-written GEMM chain runs
        return BiomedConfig(embed_dim=64, image_size=64, patch_size=16, vision=TowerConfig(256, 3, 4, 512), text=TowerConfig(256, 3, 4, 512),
                            vocab_size=500, context_length=40, max_positions=48, proj_hidden=160)
    if size == "full":
        return BiomedConfig()
    raise ValueError(size)


def hf_pair(size="tiny", seed=0, scale=0.05, ln_spread=0.2):
    """-> (hf_vit, hf_bert, ours) with identical weights; LayerNorm gains / biases are perturbed so that the fold has something to fold"""
    from transformers import BertConfig, BertModel, ViTConfig, ViTModel
    cfg = config(size)
    v, t = cfg.vision, cfg.text
    torch.manual_seed(seed)
    vit = ViTModel(ViTConfig(hidden_size=v.width, num_hidden_layers=v.layers, num_attention_heads=v.heads, intermediate_size=v.mlp,
                             image_size=cfg.image_size, patch_size=cfg.patch_size, hidden_act="gelu", layer_norm_eps=cfg.layer_norm_eps,
                             qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), add_pooling_layer=False).eval()
    bert = BertModel(BertConfig(vocab_size=cfg.vocab_size, hidden_size=t.width, num_hidden_layers=t.layers, num_attention_heads=t.heads,
                                intermediate_size=t.mlp, max_position_embeddings=cfg.max_positions, type_vocab_size=cfg.type_vocab_size,
                                hidden_act="gelu", layer_norm_eps=cfg.text_layer_norm_eps, pad_token_id=cfg.pad_token_id,
                                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in (vit, bert):
            for name, p in mod.named_parameters():
                if "LayerNorm" in name or "layernorm" in name:
                    p.copy_((1.0 if name.endswith("weight") else 0.0) + ln_spread * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * (scale if p.dim() > 1 else 0.02))
    ours = BiomedCLIP(cfg).eval()
    ours.reset_parameters(seed)                      # (projections: not part of the HF modules)
    sd = ours.open_clip_state_dict()
    hv, hb = vit.state_dict(), bert.state_dict()
    sd["visual.trunk.cls_token"] = hv["embeddings.cls_token"]
    sd["visual.trunk.pos_embed"] = hv["embeddings.position_embeddings"]
    sd["visual.trunk.patch_embed.proj.weight"] = hv["embeddings.patch_embeddings.projection.weight"]
    sd["visual.trunk.patch_embed.proj.bias"] = hv["embeddings.patch_embeddings.projection.bias"]
    sd["visual.trunk.norm.weight"], sd["visual.trunk.norm.bias"] = hv["layernorm.weight"], hv["layernorm.bias"]
    for i in range(v.layers):
        h, o = f"layers.{i}.", f"visual.trunk.blocks.{i}."
        for kind in ("weight", "bias"):
            sd[o + f"attn.qkv.{kind}"] = torch.cat([hv[h + f"attention.{x}_proj.{kind}"] for x in "qkv"], 0)
            sd[o + f"attn.proj.{kind}"] = hv[h + f"attention.o_proj.{kind}"]
            sd[o + f"norm1.{kind}"], sd[o + f"norm2.{kind}"] = hv[h + f"layernorm_before.{kind}"], hv[h + f"layernorm_after.{kind}"]
            sd[o + f"mlp.fc1.{kind}"], sd[o + f"mlp.fc2.{kind}"] = hv[h + f"mlp.fc1.{kind}"], hv[h + f"mlp.fc2.{kind}"]
    for k, val in hb.items():                        # open_clip keeps the HF module under `text.transformer.`
        sd["text.transformer." + k] = val
    ours.load_open_clip_state_dict(sd)
    return vit, bert, ours


@torch.no_grad()
def hf_image_features(vit, ours, px):
    return vit(pixel_values=px).last_hidden_state[:, 0] @ ours.vision.proj.weight.T


@torch.no_grad()
def hf_text_features(bert, ours, ids):
    h = bert(input_ids=ids, attention_mask=(ids != ours.cfg.pad_token_id).long()).last_hidden_state[:, 0]
    return torch.nn.functional.gelu(h @ ours.text.proj1.weight.T) @ ours.text.proj2.weight.T


def caption_ids(cfg, lengths, seed=0):
    """[CLS]=2 ... [SEP]=3, zero padded, the given token counts (each >= 2)"""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(len(lengths), cfg.context_length, dtype=torch.long)
    for r, L in enumerate(lengths):
        ids[r, 0], ids[r, L - 1] = 2, 3
        ids[r, 1:L - 1] = torch.randint(4, cfg.vocab_size, (L - 2,), generator=g)
    return ids
