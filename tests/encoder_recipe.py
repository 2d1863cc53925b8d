"""Seeded recipe shared by tools/make_golden_encoder.py (build container, reference side) and the encoder tests
(both sides regenerate the SAME weights and inputs from seeds, so only the reference's OUTPUTS are stored)."""
import zlib

import torch

CONFIGS = {
    # name: kwargs of lib/models/chexzero_clip.py CLIP(...)  (:263-277)
    "small_hd64": dict(embed_dim=64, image_resolution=32, vision_layers=2, vision_width=128, vision_patch_size=8,
                       context_length=16, vocab_size=300, transformer_width=64, transformer_heads=1, transformer_layers=2),
    # load_clip(None, 77) (:458-470): the architecture of the cc3m_clip_from_scratch branch
    "scratch_b16_77": dict(embed_dim=768, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=16,
                           context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                           transformer_layers=12),
    # load_clip(None) with its DEFAULT context_length = 256 (:458): the mimic_clip_from_scratch_* / chexzero branches
    # (lib/models/utils.py:79-95): 256-token causal text tower
    "scratch_b16_256": dict(embed_dim=768, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=16,
                            context_length=256, vocab_size=49408, transformer_width=512, transformer_heads=8,
                            transformer_layers=12),
}


def _gen(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)


def openai_state_dict(cfg, seed=0):
    """Every tensor of an OpenAI-format CLIP state dict (chexzero_clip.py:263-392) from its own seeded generator:
    matrices N(0, 1/sqrt(fan_in)), biases N(0, 0.02), LayerNorm weights 1 + N(0, 0.1), embeddings N(0, 0.02)."""
    vw, tw, E = cfg["vision_width"], cfg["transformer_width"], cfg["embed_dim"]
    P, grid = cfg["vision_patch_size"], cfg["image_resolution"] // cfg["vision_patch_size"]
    sd = {}

    def t(name, shape, std):
        sd[name] = torch.randn(shape, generator=_gen(name, seed)) * std

    def ln(name, w):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(w, generator=_gen(name + ".weight", seed))
        t(name + ".bias", (w,), 0.02)

    def blocks(prefix, w, n):
        for i in range(n):
            p = f"{prefix}transformer.resblocks.{i}."
            t(p + "attn.in_proj_weight", (3 * w, w), w ** -0.5); t(p + "attn.in_proj_bias", (3 * w,), 0.02)
            t(p + "attn.out_proj.weight", (w, w), w ** -0.5); t(p + "attn.out_proj.bias", (w,), 0.02)
            ln(p + "ln_1", w); ln(p + "ln_2", w)
            t(p + "mlp.c_fc.weight", (4 * w, w), w ** -0.5); t(p + "mlp.c_fc.bias", (4 * w,), 0.02)
            t(p + "mlp.c_proj.weight", (w, 4 * w), (4 * w) ** -0.5); t(p + "mlp.c_proj.bias", (w,), 0.02)

    t("visual.conv1.weight", (vw, 3, P, P), (3 * P * P) ** -0.5)
    t("visual.class_embedding", (vw,), vw ** -0.5)
    t("visual.positional_embedding", (grid * grid + 1, vw), vw ** -0.5)
    ln("visual.ln_pre", vw); ln("visual.ln_post", vw)
    blocks("visual.", vw, cfg["vision_layers"])
    t("visual.proj", (vw, E), vw ** -0.5)
    t("token_embedding.weight", (cfg["vocab_size"], tw), 0.02)
    t("positional_embedding", (cfg["context_length"], tw), 0.01)
    blocks("", tw, cfg["transformer_layers"])
    ln("ln_final", tw)
    t("text_projection", (tw, E), tw ** -0.5)
    sd["logit_scale"] = torch.tensor(2.6592)
    return sd


def inputs(cfg, seed=0, n_img=2, n_txt=5):
    """pixel_values [n_img,3,S,S] ~ N(0,1); token ids [n_txt, ctx] with ragged lengths, EOT = vocab-1 (largest id),
    zero padded -- one full-length row and one 3-token row included."""
    S, ctx, V = cfg["image_resolution"], cfg["context_length"], cfg["vocab_size"]
    px = torch.randn((n_img, 3, S, S), generator=_gen("pixels", seed))
    g = _gen("tokens", seed)
    ids = torch.randint(1, V - 2, (n_txt, ctx), generator=g)
    lens = [ctx, 3] + [int(v) for v in torch.randint(4, ctx, (max(n_txt - 2, 0),), generator=g)]
    for i, L in enumerate(lens[:n_txt]):
        ids[i, 0] = V - 2
        ids[i, L - 1] = V - 1
        ids[i, L:] = 0
    return px, ids


def lemon_clip_from_recipe(name, seed=0):
    from lemon_amd.clip import ClipConfig, LemonCLIP
    sd = openai_state_dict(CONFIGS[name], seed)
    return LemonCLIP(ClipConfig.from_openai_state_dict(sd)).load_openai_state_dict(sd).eval()
