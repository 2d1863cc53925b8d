"""GPU: the fused encoder path (lemon_linear_f32 epilogues, lemon_attention_f32, pooled-row last block, patch-embedding
GEMM, QuickGELU composed from SiLU) at REAL widths against reference-side modules on the CPU:
  * HF transformers CLIPModel -- the class the reference wraps (lib/models/downstream_models.py:30-41) -- with identical
    seeded weights, for the three architectures BASELINE.json names (ViT-B/32, ViT-B/16, ViT-L/14);
  * the reference's in-tree CLIP (lib/models/chexzero_clip.py) through tests/golden/encoder_chexzero.npz;
  * tier B: CPU-HF embeddings -> oracle scores vs GPU embeddings -> HIP scores, AUROC to 3 decimals."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ARCH = {
    "vit-b-32": dict(projection_dim=512, v=(768, 12, 12, 3072, 32), t=(512, 12, 8, 2048)),
    "vit-b-16": dict(projection_dim=512, v=(768, 12, 12, 3072, 16), t=(512, 12, 8, 2048)),
    "vit-l-14": dict(projection_dim=768, v=(1024, 24, 16, 4096, 14), t=(768, 12, 12, 3072)),
}


def hf_model(arch, seed=0):
    from transformers import CLIPConfig, CLIPModel
    a = ARCH[arch]
    cfg = CLIPConfig(projection_dim=a["projection_dim"],
                     vision_config=dict(hidden_size=a["v"][0], num_hidden_layers=a["v"][1], num_attention_heads=a["v"][2],
                                        intermediate_size=a["v"][3], image_size=224, patch_size=a["v"][4]),
                     text_config=dict(hidden_size=a["t"][0], num_hidden_layers=a["t"][1], num_attention_heads=a["t"][2],
                                      intermediate_size=a["t"][3], vocab_size=49408, max_position_embeddings=77,
                                      eos_token_id=2, bos_token_id=0, pad_token_id=1))     # legacy ids => argmax EOT pooling
    torch.manual_seed(seed)
    hf = CLIPModel(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in hf.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (0.02 if "embedding" in name else fan_in ** -0.5))
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return hf


def _unwrap(o):
    return o if torch.is_tensor(o) else o.pooler_output


def ragged_ids(n, ctx, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, vocab - 2, (n, ctx), generator=g)
    lens = [ctx, 3, 8, 9] + [int(v) for v in torch.randint(4, ctx, (max(n - 4, 0),), generator=g)]
    mask = torch.zeros(n, ctx, dtype=torch.long)
    for i, L in enumerate(lens[:n]):
        ids[i, 0] = vocab - 2
        ids[i, L - 1] = vocab - 1
        ids[i, L:] = 0
        mask[i, :L] = 1
    return ids, mask


@pytest.mark.parametrize("gemm", ["split", "f16x3", "f32"])
@pytest.mark.parametrize("arch", ["vit-b-32", "vit-b-16", "vit-l-14"])
def test_fused_gpu_encoder_vs_hf_clip_at_full_size(hip, arch, gemm, monkeypatch):
    # gemm = "split" (the default): QKV / output projection / fc1 as 3-way bf16 split GEMMs (lemon_linear_bf16x6);
    # "f32": every GEMM on the fp32 matrix cores.  The SAME bars for both.
    monkeypatch.setenv("LEMON_GEMM", gemm)
    from lemon_amd.clip import ClipConfig, LemonCLIP
    from lemon_amd.data import gpu_transform_batch
    from lemon_amd.ops import normalize_vectors
    hf = hf_model(arch)
    cfg = ClipConfig.named(arch)
    ours = LemonCLIP(cfg).load_hf_state_dict(hf.state_dict()).eval().cuda()
    n_img = 3 if arch == "vit-l-14" else 6
    u8 = torch.randint(0, 256, (n_img, 48, 40, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).cuda()
    px = gpu_transform_batch(u8, 224)                                  # NCHW float (bit-identical to PIL, tested elsewhere)
    patches = gpu_transform_batch(u8, 224, patch=cfg.patch_size)       # patch-major: the bench / run_lemon input form
    ids, mask = ragged_ids(7, 77, 49408, seed=3)
    with torch.no_grad():
        ref_img = _unwrap(hf.get_image_features(pixel_values=px.cpu()))
        ref_txt = _unwrap(hf.get_text_features(input_ids=ids, attention_mask=mask))
    got_nchw = ours.encode_image(px).cpu()
    got_patch = ours.encode_image(patches).cpu()
    got_txt = ours.encode_text(ids.cuda(), mask.cuda()).cpu()
    scale_i, scale_t = float(ref_img.abs().max()), float(ref_txt.abs().max())
    for name, got, ref, sc in (("nchw", got_nchw, ref_img, scale_i), ("patch-major", got_patch, ref_img, scale_i),
                               ("text", got_txt, ref_txt, scale_t)):
        d = float((got - ref).abs().max())
        assert d <= 1e-4 * max(1.0, sc), f"{arch}/{name}: raw max abs diff {d} (scale {sc})"
        dn = float((torch.nn.functional.normalize(got, dim=1) - torch.nn.functional.normalize(ref, dim=1)).abs().max())
        assert dn <= 5e-6, f"{arch}/{name}: normalised max abs diff {dn}"
    # the product's own normalisation kernel on the GPU embeddings
    nn_ = normalize_vectors(ours.encode_text(ids.cuda())).cpu()
    assert float((nn_ - torch.nn.functional.normalize(ref_txt, dim=1)).abs().max()) <= 5e-6


@pytest.mark.parametrize("name", ["small_hd64", "scratch_b16_77", "scratch_b16_256"])
def test_fused_gpu_encoder_vs_reference_in_tree_clip(hip, name):
    from tests.encoder_recipe import CONFIGS, inputs, lemon_clip_from_recipe
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_chexzero.npz"))
    m = lemon_clip_from_recipe(name).cuda()
    px, ids = inputs(CONFIGS[name])
    gi, gt = m.encode_image(px.cuda()).cpu().numpy(), m.encode_text(ids.cuda()).cpu().numpy()
    ri, rt = fx[f"{name}_img"], fx[f"{name}_txt"]
    assert np.abs(gi - ri).max() <= 1e-4 * max(1.0, np.abs(ri).max()), np.abs(gi - ri).max()
    assert np.abs(gt - rt).max() <= 1e-4 * max(1.0, np.abs(rt).max()), np.abs(gt - rt).max()


def test_tier_b_cpu_hf_embeddings_vs_gpu_pipeline(hip, oracle):
    """2 048 planted CIFAR-shaped samples: HF CLIPModel (CPU) embeddings -> oracle neighbours/scores, against the same
    images and prompts through the GPU pipeline (preprocess + fused encoder + HIP kNN/score).  Reports max |dscore| and
    the fraction of samples whose neighbour set changed; AUROC must agree to 3 decimals for d_1 alone and for the fixed
    hyper-parameters (SURVEY 8d metric (i), (ii))."""
    from lemon_amd import datasets as ds
    from lemon_amd.clip import ClipConfig, LemonCLIP, SyntheticTokenizer
    from lemon_amd.data import gpu_transform_batch
    from lemon_amd.pipeline import Embedder, FIXED_HPARAMS, run_hot_path
    arch = "vit-b-32"
    hf = hf_model(arch, seed=4)
    cfg = ClipConfig.named(arch)
    ours = LemonCLIP(cfg).load_hf_state_dict(hf.state_dict())
    dev = torch.device("cuda", 0)
    n_tr, n_q, C, k = 1792, 256, 10, 5
    rng = np.random.default_rng(0)
    pat = rng.integers(0, 256, (C, 32, 32, 3)).astype(np.int16)
    tok = SyntheticTokenizer(cfg.vocab_size, cfg.context_length, cfg.eos_token_id)
    class_ids = torch.tensor(tok(["A photo of a " + l for l in ds.cifar10_labels])["input_ids"])
    data, clean_noisy = {}, {}
    for name, n in (("train", n_tr), ("val", n_q)):
        clean = rng.integers(0, C, n)
        noisy = np.where(rng.random(n) < 0.4, (clean + 1) % C, clean)
        u8 = np.clip(pat[clean] + rng.integers(-48, 49, (n, 32, 32, 3)), 0, 255).astype(np.uint8)
        data[name] = dict(pixels=torch.from_numpy(u8).to(dev), ids=class_ids[torch.from_numpy(noisy)].to(dev),
                          label_id=torch.from_numpy(noisy.astype(np.int32)).to(dev))
        clean_noisy[name] = (clean, noisy)
    # text_dedup: each distinct prompt is embedded once and gathered, as the CPU side below does.  (Without it the same
    # prompt embedded in micro-batches of different row counts can differ in the last bit -- another GEMM solution --
    # which changes WHICH of the tied text-side neighbours are picked; the reference's batch-of-128 loop has the
    # same property.)
    emb = Embedder(ours, dev, batch_size=256, text_dedup=True)
    recs, db = run_hot_path(emb, data, k=k, dist_type="cosine", hparams=FIXED_HPARAMS)
    torch.cuda.synchronize()

    # reference side: HF on the CPU, same pixels (the GPU transform is bit-identical to PIL + torch), same prompts
    ref = {}
    with torch.no_grad():
        cls_txt = _unwrap(hf.get_text_features(input_ids=class_ids, attention_mask=(class_ids != 0).long()))
        for name in ("train", "val"):
            outs = []
            for i in range(0, data[name]["pixels"].shape[0], 128):
                px = gpu_transform_batch(data[name]["pixels"][i:i + 128], 224).cpu()
                outs.append(_unwrap(hf.get_image_features(pixel_values=px)))
            img = oracle.normalize_rows(torch.cat(outs).numpy())
            txt = oracle.normalize_rows(cls_txt[torch.from_numpy(clean_noisy[name][1])].numpy())
            ref[name] = (img, txt)
    out = oracle.neighbors("cosine", ref["train"][0], ref["train"][1], ref["val"][0], ref["val"][1], k)
    s_ref = oracle.score(out, FIXED_HPARAMS)
    s_gpu = recs["val"]["score"].cpu().numpy()
    y = clean_noisy["val"][0] != clean_noisy["val"][1]
    d_emb = np.abs(recs["val"]["emb_img"].cpu().numpy() - ref["val"][0]).max()
    ch_n = (np.sort(recs["val"]["I_n"].cpu().numpy(), 1) != np.sort(out["I_n"], 1)).any(1)
    ch_m = (np.sort(recs["val"]["I_m"].cpu().numpy(), 1) != np.sort(out["I_m"], 1)).any(1)
    same = ~(ch_n | ch_m)
    dscore = float(np.abs(s_gpu - s_ref).max())
    dscore_same = float(np.abs(s_gpu - s_ref)[same].max())
    print(f"tier-B: max|d emb|={d_emb:.2e}  max|d score|={dscore:.2e} (same neighbour sets: {dscore_same:.2e})  "
          f"neighbour sets changed: image {ch_n.mean():.4f} text {ch_m.mean():.4f}")
    assert d_emb <= 5e-6
    assert dscore_same <= 1e-4                       # north_star: scores within 1e-4 where the index sets agree
    assert ch_n.mean() <= 0.02 and ch_m.mean() <= 0.02
    a = lambda s: round(oracle.auroc(y, s), 3)
    assert a(recs["val"]["d_1"].cpu().numpy()) == a(out["d_1"])
    assert a(s_gpu) == a(s_ref)
    # planted structure: the check is informative (a random-init text tower separates class prompts only weakly, so
    # the fixed-hparam AUROC is modest; with the neighbour terms weighted up it is clearly above chance)
    strong = dict(FIXED_HPARAMS, beta=100.0, gamma=100.0)
    a_ref, a_gpu = oracle.auroc(y, oracle.score(out, strong)), oracle.auroc(y, oracle.score({k_: v.cpu().numpy() for k_, v in recs["val"].items() if k_ != "score"}, strong))
    print(f"tier-B AUROC: fixed hparams {oracle.auroc(y, s_ref):.4f}; beta=gamma=100: oracle {a_ref:.4f} gpu {a_gpu:.4f}")
    assert round(a_ref, 3) == round(a_gpu, 3) and a_ref > 0.6 and oracle.auroc(y, s_ref) > 0.55


def test_f16x3_reembeds_micro_batches_beyond_the_fp16_range(hip, monkeypatch):
    # LEMON_GEMM=f16x3 carries the block GEMMs' fp32 operands as fp16 pairs: an activation beyond +-65 504 is never clamped into a
    # silently different embedding -- the micro-batch is embedded again with range-free operands (bf16x6 GEMMs, fp32 attention) and
    # counted; with the fallback off it ends in an error.  bf16x6 / f32 have no such limit.
    from lemon_amd.clip import ClipConfig, LemonCLIP
    from lemon_amd.pipeline import Embedder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = LemonCLIP(ClipConfig.named("tiny")).eval()
    px = torch.randn(8, 3, 32, 32).to(dev)
    outs = {}
    for scale in (1.0, 3.0e6):
        with torch.no_grad():
            model.vision.blocks[0].ln1.weight.fill_(scale)           # LayerNorm output ~ scale: 3e6 is beyond fp16
        for mode in ("f32", "bf16x6", "f16x3"):
            monkeypatch.setenv("LEMON_GEMM", mode)
            emb = Embedder(model, dev, batch_size=4)
            e = emb.embed_images(px)
            emb.raise_if_nonfinite()
            assert bool(torch.isfinite(e).all())
            assert emb.fallback_batches == (2 if (mode == "f16x3" and scale > 1.0) else 0), (mode, scale, emb.fallback_batches)
            outs[(scale, mode)] = e
        assert (outs[(scale, "bf16x6")] - outs[(scale, "f32")]).abs().max() < 5e-6      # (unit-norm embeddings)
        assert (outs[(scale, "f16x3")] - outs[(scale, "f32")]).abs().max() < 5e-6
    assert torch.equal(outs[(3.0e6, "f16x3")], outs[(3.0e6, "bf16x6")])                 # the re-embedded batches ARE bf16x6 batches
    monkeypatch.setenv("LEMON_GEMM", "f16x3")
    emb = Embedder(model, dev, batch_size=4, range_fallback=False)
    emb.embed_images(px)
    with pytest.raises(FloatingPointError, match="fp16 range"):
        emb.raise_if_nonfinite()


def test_folded_layernorm_reembeds_micro_batches_whose_rows_are_far_from_zero_mean(hip, monkeypatch):
    # LEMON_LNFOLD (default on with f16x3 / block): the folded GEMM's error grows with |mean| / sigma of a LayerNorm input row; rows
    # beyond ops.LN_FOLD_MAX_SHIFT are poisoned by the kernels and their micro-batch is embedded again with LayerNorm kernels
    # (same f16x3 arithmetic; bf16x6 only if that is still not finite) -- never a silently less accurate embedding.  A pre-LayerNorm bias of 40 puts every token row of block 0
    # at mean 40, sigma ~ 1.
    from lemon_amd import ops
    from lemon_amd.clip import ClipConfig, LemonCLIP
    from lemon_amd.pipeline import Embedder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = LemonCLIP(ClipConfig.named("vit-b-32")).eval()
    px = torch.randn(6, 3, 224, 224).to(dev)
    outs = {}
    for shift in (0.0, 40.0):
        with torch.no_grad():
            model.vision.pre_ln.bias.fill_(shift)
        for mode in ("f32", "f16x3"):
            monkeypatch.setenv("LEMON_GEMM", mode)
            emb = Embedder(model, dev, batch_size=3)
            e = emb.embed_images(px)
            emb.raise_if_nonfinite()
            assert bool(torch.isfinite(e).all())
            assert emb.fold_fallback_batches == (2 if (mode == "f16x3" and shift > ops.LN_FOLD_MAX_SHIFT) else 0), (mode, shift, emb.fold_fallback_batches)
            assert emb.fallback_batches == 0      # (nothing left the fp16 range: the second stage, bf16x6, is not needed)
            outs[(shift, mode)] = e
        assert (outs[(shift, "f16x3")] - outs[(shift, "f32")]).abs().max() < 5e-6          # (unit-norm embeddings)
    monkeypatch.setenv("LEMON_GEMM", "f16x3")
    monkeypatch.setenv("LEMON_LNFOLD", "0")                                                   # LayerNorm kernels: no bound, no fallback
    emb = Embedder(model, dev, batch_size=3)
    e = emb.embed_images(px)
    assert emb.fallback_batches == 0 and emb.fold_fallback_batches == 0 and (e - outs[(40.0, "f32")]).abs().max() < 5e-6


def test_only_the_flagged_samples_are_embedded_again(hip, monkeypatch):
    # round 4 re-embedded the whole micro-batch of a flagged row (5 240 images at the headline shape); samples are independent, so
    # now only the non-finite SAMPLES are gathered into a sub-batch and embedded again.  Planted here:
    # (a) three captions that carry a token whose embedding row sits at mean 40, sigma 0.5: beyond the folded LayerNorm's bound in
    #     block 0 (NaN row affine from the kernels) -> repaired by the first stage (LayerNorm kernels, same f16x3 arithmetic);
    # (b) three images whose embeddings come back non-finite from every f16x3 pass (a wrapper poisons them: an activation beyond the
    #     fp16 range cannot be planted per image behind CLIP's pre-LayerNorm; the whole-batch case is the test above) -> both stages
    #     run on exactly those three images, the second (bf16x6) repairs them.
    from lemon_amd import ops
    from lemon_amd.clip import ClipConfig, LemonCLIP
    from lemon_amd.pipeline import Embedder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = LemonCLIP(ClipConfig.named("vit-b-32")).eval()
    n, bs = 40, 16
    px = torch.randn(n, 3, 224, 224)
    planted_img = [3, 17, 38]                      # micro-batches 0, 1, 2
    px[planted_img, 0, 0, 0] = 12345.0             # (the wrapper's mark)
    ids = torch.randint(1, 1000, (n, 77))
    ids[:, 9] = model.cfg.eos_token_id
    special = 1234
    planted_txt = [0, 1, 33]                       # micro-batches 0 and 2 of 16
    ids[planted_txt, 4] = special
    with torch.no_grad():
        model.text.tok.weight[special] = 40.0 + 0.5 * torch.randn(model.cfg.text.width)
    calls = []
    real_img, real_txt = model.encode_image, model.encode_text

    def enc_img(x, *a, **k):
        calls.append(("img", int(x.shape[0])))
        y = real_img(x, *a, **k)
        if ops.gemm_mode() == "f16x3":
            y = torch.where((x[:, 0, 0, 0] == 12345.0)[:, None], torch.full_like(y, float("nan")), y)
        return y

    monkeypatch.setattr(model, "encode_image", enc_img)
    monkeypatch.setattr(model, "encode_text", lambda x, *a, **k: (calls.append(("txt", int(x.shape[0]))), real_txt(x, *a, **k))[1])
    outs = {}
    for mode in ("f32", "f16x3"):
        monkeypatch.setenv("LEMON_GEMM", mode)
        calls.clear()
        emb = Embedder(model, dev, batch_size=bs, text_batch_size=bs)
        ei, et = emb.embed_images(px.to(dev)), emb.embed_texts(ids.to(dev))
        emb.raise_if_nonfinite()
        assert bool(torch.isfinite(ei).all()) and bool(torch.isfinite(et).all())
        outs[mode] = (ei, et)
        if mode == "f32":
            assert (emb.fallback_rows, emb.fold_fallback_rows) == (0, 0) and len(calls) == 6
        else:
            # every extra encoder call is a sub-batch of the planted samples, never a micro-batch
            assert emb.fallback_rows == 3 and emb.fallback_batches == 3, (emb.fallback_rows, emb.fallback_batches)
            assert emb.fold_fallback_rows == 3 and emb.fold_fallback_batches == 2, (emb.fold_fallback_rows, emb.fold_fallback_batches)
            assert calls[:3] == [("img", 16), ("img", 16), ("img", 8)] and calls[5:8] == [("txt", 16), ("txt", 16), ("txt", 8)], calls
            assert calls[3:5] + calls[8:] == [("img", 3), ("img", 3), ("txt", 3)], calls
    for a, b in zip(outs["f16x3"], outs["f32"]):
        assert float((a - b).abs().max()) < 5e-6                       # (unit-norm embeddings)


@pytest.mark.parametrize("L,H,causal", [(50, 12, False), (24, 8, True), (197, 12, False)])
def test_attention_range_and_small_magnitudes_on_head_dim_64(hip, L, H, causal):
    # the three attention kernels (one / two key tiles: short kernel; L > 64: general kernel) at head_dim 64, both arithmetic
    # forms, against float64: (a) O(1) inputs, (b) q, k of magnitude 1e-3 (the general kernel's UNscaled lo parts keep 2^-25
    # absolute), (c) v of magnitude 1e4 .. 6e4 (inside fp16) and (d) beyond the fp16 range: the fp16 form must go non-finite (loud), the
    # fp32 form (lemon_attention_set_f16(0), what the f32 / bf16x6 GEMM modes select) must stay exact
    from lemon_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(L * 31 + H)
    B, W = 3, 64 * H

    def ref(qkv):
        q, k, v = qkv.double().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        s = q @ k.transpose(-1, -2) / 8.0
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
        return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, W)

    cases = {"unit": (1.0, 1.0), "small_qk": (1e-3, 1.0), "large_v": (1.0, 1.0e4), "beyond_fp16": (1.0, 2.0e5)}
    prev = lib.lemon_attention_set_f16(1)
    try:
        for name, (sq, sv) in cases.items():
            qkv = torch.randn(B, L, 3 * W, generator=g)
            qkv.view(B, L, 3, W)[:, :, :2] *= sq
            qkv.view(B, L, 3, W)[:, :, 2] *= sv
            if name == "large_v":
                qkv.clamp_(-6.0e4, 6.0e4)                      # every value inside the fp16 range (65 504)
            want = ref(qkv)
            tol = 3e-6 * float(want.abs().max()) + 1e-12
            for f16 in (1, 0):
                lib.lemon_attention_set_f16(f16)
                got = ops.attention(qkv.cuda(), H, causal).cpu().double()
                if name == "beyond_fp16" and f16:
                    assert not bool(torch.isfinite(got).all()), "fp16 attention must not clamp out-of-range inputs"
                else:
                    assert bool(torch.isfinite(got).all()) and float((got - want).abs().max()) <= tol, (name, f16, float((got - want).abs().max()), tol)
    finally:
        lib.lemon_attention_set_f16(prev)
        ops._attn_f16_state = None


def test_fused_mlp_and_library_mlp_give_the_same_embeddings(hip, monkeypatch):
    # LEMON_MLP=block (all four GEMMs of a block in gemm_f16x3.hip, the default) / fused (the MLP only) vs lib (hipBLASLt + split
    # pass), ViT-B/32, both towers
    from lemon_amd.clip import ClipConfig, LemonCLIP
    from lemon_amd.ops import normalize_vectors
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = LemonCLIP(ClipConfig.named("vit-b-32")).eval().to(dev)
    px = torch.randn(70, 3, 224, 224, device=dev)
    ids = torch.randint(1, 1000, (300, 77), device=dev)
    ids[:, 9] = model.cfg.eos_token_id
    out = {}
    for mode in ("block", "fused", "lib"):
        monkeypatch.setenv("LEMON_MLP", mode)
        with torch.no_grad():
            out[mode] = (normalize_vectors(model.encode_image(px).float()), normalize_vectors(model.encode_text(ids).float()))
    for mode in ("block", "fused"):
        for a, b in zip(out[mode], out["lib"]):
            assert bool(torch.isfinite(a).all()) and float((a - b).abs().max()) < 5e-6, mode
