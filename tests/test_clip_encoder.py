"""Encoder parity (CPU): our CLIP module vs HF transformers' CLIPModel on identical random weights.
HF CLIPModel is the class the reference wraps (lib/models/downstream_models.py:30-41)."""
import numpy as np
import pytest
import torch

from lemon_amd.clip import ClipConfig, LemonCLIP, SyntheticTokenizer, algorithm_class_from_scratch, encoder_flops


def _hf_pair(seed=0):
    transformers = pytest.importorskip("transformers")
    from transformers import CLIPConfig, CLIPModel
    cfg = CLIPConfig(
        projection_dim=32,
        vision_config=dict(hidden_size=48, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96,
                           image_size=32, patch_size=8),
        text_config=dict(hidden_size=40, num_hidden_layers=2, num_attention_heads=4, intermediate_size=80,
                         vocab_size=300, max_position_embeddings=16, eos_token_id=2, bos_token_id=0, pad_token_id=1))
    torch.manual_seed(seed)
    hf = CLIPModel(cfg).eval()
    with torch.no_grad():
        for p in hf.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    ours = LemonCLIP(ClipConfig.named("tiny")).eval()
    ours.load_hf_state_dict(hf.state_dict())
    return hf, ours


def _unwrap(o):
    return o if torch.is_tensor(o) else o.pooler_output


def test_matches_hf_clip_on_random_weights():
    hf, ours = _hf_pair()
    g = torch.Generator().manual_seed(1)
    px = torch.randn(5, 3, 32, 32, generator=g)
    ids = torch.randint(3, 298, (5, 16), generator=g)
    lens = [4, 9, 16, 7, 12]
    mask = torch.zeros(5, 16, dtype=torch.long)
    for i, L in enumerate(lens):
        ids[i, L - 1] = 299           # EOT = largest id
        ids[i, L:] = 0
        mask[i, :L] = 1
    with torch.no_grad():
        ref_img = _unwrap(hf.get_image_features(pixel_values=px))
        ref_txt = _unwrap(hf.get_text_features(input_ids=ids, attention_mask=mask))
    got_img, got_txt = ours.encode_image(px), ours.encode_text(ids, mask)
    assert got_img.shape == (5, 32) and got_txt.shape == (5, 32)
    assert (got_img - ref_img).abs().max() < 2e-5, (got_img - ref_img).abs().max()
    assert (got_txt - ref_txt).abs().max() < 2e-5, (got_txt - ref_txt).abs().max()


def test_text_dedup_and_truncation_are_exact():
    _, ours = _hf_pair(3)
    tok = SyntheticTokenizer(300, 16, 299)
    prompts = ["A photo of a cat", "A photo of a dog", "A photo of a cat", "A photo of a streetcar"] * 3
    enc = tok(prompts, padding="max_length", truncation=True)
    ids = torch.tensor(enc["input_ids"])
    assert ids.shape == (12, 16) and (ids.argmax(-1) == torch.tensor(enc["attention_mask"]).sum(-1) - 1).all()
    full = ours.encode_text(ids)
    assert torch.equal(full[0], full[2])
    assert (ours.encode_text_dedup(ids) - full).abs().max() < 1e-6


def test_factory_surface():
    model, tok = algorithm_class_from_scratch("huggingface_clip", "random:tiny", None, return_tokenizer=True)
    assert hasattr(model, "encode_text") and hasattr(model, "encode_image")
    out = tok(["a b c"], padding="max_length", truncation=True)
    assert set(out) >= {"input_ids", "attention_mask"} and len(out["input_ids"][0]) == 16
    with pytest.raises(FileNotFoundError):                 # (the biomed_clip branch: tests/test_biomed.py)
        algorithm_class_from_scratch("biomed_clip", "x", None)
    with pytest.raises(NotImplementedError):
        algorithm_class_from_scratch("medclip", "x", None)
    with pytest.raises(FileNotFoundError):
        algorithm_class_from_scratch("huggingface_clip", "openai/clip-vit-base-patch32", None)


def test_flop_model_matches_survey():
    img, txt = encoder_flops(ClipConfig.named("vit-b-32"))
    assert 8.0e9 < img < 9.6e9 and 5.4e9 < txt < 6.6e9          # ~8.8 / ~6.0 GFLOP
    n = sum(p.numel() for p in LemonCLIP(ClipConfig.named("vit-b-32")).parameters())
    assert abs(n - 151.3e6) < 0.3e6                              # 151.3 M params


# ------------------------------------------------------------------ in-tree CLIP (A3'): reference-generated goldens
@pytest.mark.parametrize("name", ["small_hd64", "scratch_b16_77", "scratch_b16_256"])
def test_openai_format_loader_matches_reference_clip(name):
    """LemonCLIP loaded from an OpenAI-format state dict vs the outputs of the reference's own CLIP module
    (lib/models/chexzero_clip.py:263-392; `scratch_b16_77` is built by its load_clip(None, 77), :458-479) on the same
    seeded weights and inputs (tools/make_golden_encoder.py -> tests/golden/encoder_chexzero.npz)."""
    import os
    from tests.encoder_recipe import CONFIGS, inputs, lemon_clip_from_recipe, openai_state_dict
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_chexzero.npz"))
    cfg = CONFIGS[name]
    sd = openai_state_dict(cfg)
    px, ids = inputs(cfg)
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) / float(fx[f"{name}_weights_abs_sum"]) - 1) < 1e-9
    assert abs((float(px.double().abs().sum()) + float(ids.double().sum())) / float(fx[f"{name}_inputs_abs_sum"]) - 1) < 1e-9
    m = lemon_clip_from_recipe(name)
    assert m.context_length == cfg["context_length"] and m.cfg.embed_dim == cfg["embed_dim"]
    gi, gt = m.encode_image(px).numpy(), m.encode_text(ids).numpy()
    ri, rt = fx[f"{name}_img"], fx[f"{name}_txt"]
    assert np.abs(gi - ri).max() < 2e-5 * max(1.0, np.abs(ri).max()), np.abs(gi - ri).max()
    assert np.abs(gt - rt).max() < 2e-5 * max(1.0, np.abs(rt).max()), np.abs(gt - rt).max()


def test_in_tree_branches_of_the_factory(tmp_path):
    """algorithm_class_from_scratch for the in-tree CLIP branches (lib/models/utils.py:82-103): tokenizer -> LongTensor,
    encode_text(tokens); a checkpoint of the wrong architecture is refused; an unknown branch is refused."""
    from tests.encoder_recipe import CONFIGS, openai_state_dict
    model, tok = algorithm_class_from_scratch("cc3m_clip_from_scratch", "random:tiny", None, return_tokenizer=True)
    t = tok(["a b c", "d"])
    assert t.dtype == torch.long and t.shape == (2, 16)
    assert model.encode_text(t).shape == (2, 32)
    path = tmp_path / "ckpt.pt"
    torch.save(openai_state_dict(CONFIGS["small_hd64"]), path)
    m2 = LemonCLIP.from_openai_checkpoint(str(path))
    assert m2.cfg.embed_dim == 64 and m2.cfg.vision.heads == 2
    with pytest.raises(ValueError):
        algorithm_class_from_scratch("chexzero", str(path), None)
    with pytest.raises(NotImplementedError):
        algorithm_class_from_scratch("finetune", "x", None)
