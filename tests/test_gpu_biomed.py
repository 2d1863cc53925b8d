"""GPU: the `biomed_clip` branch (lib/models/utils.py:72-78; lemon_amd/biomed.py) -- timm-style ViT + BERT text tower on the HIP
library -- against HF transformers' ViTModel / BertModel on the CPU with identical seeded weights (tests/biomed_recipe.py), in
every GEMM mode; the exact-GELU epilogues and the LayerNorm-free token assembly against float64; the exact-length caption
grouping of pipeline.Embedder; the CLI on the synthetic mimic-cxr stand-in."""
import os
import pickle

import numpy as np
import pytest
import torch

from .biomed_recipe import caption_ids, hf_image_features, hf_pair, hf_text_features

pytestmark = pytest.mark.gpu


def _gelu64(x):
    return 0.5 * x * (1.0 + torch.erf(x / 2.0 ** 0.5))


def test_gelu_epilogues_against_float64(hip):
    from lemon_amd import ops
    g = torch.Generator().manual_seed(0)
    m, k, n = 300, 256, 512
    x = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).cuda()
    b = (torch.randn(n, generator=g) * 0.5).cuda()
    ref = _gelu64(x.double() @ w.double().T + b.double())
    scale = float(ref.abs().max())
    # library GEMMs: bias epilogue + the in-place pass
    for mode in ("f32", "bf16x6", "f16x3"):
        if mode == "f32":
            y = ops.linear(x, w, b, act="gelu")
        else:
            ws = ops.weight_scale_f16x3(w) if mode == "f16x3" else 1.0
            y = ops.linear_split(ops.split_operand(x, mode), ops.split_operand(w, mode, weight=True, wscale=ws), b, act="gelu", alpha=1.0 / ws)
        assert float((y.double() - ref).abs().max()) <= 4e-6 * scale, mode
    # odd sizes: the scalar tail of the in-place pass
    y = ops.linear(x[:7, :], w[:5], b[:5], act="gelu")
    assert float((y.double() - ref[:7, :5]).abs().max()) <= 4e-6 * scale
    # hand-written GEMM: the operand epilogue (plain and with a folded LayerNorm), both matrix-instruction shapes
    ws = ops.weight_scale_f16x3(w)
    wt = ops.pack_weight_t(w, ws)
    lib = __import__("lemon_amd._lib", fromlist=["load"]).load()
    for shape in (16, 32):
        prev = lib.lemon_linear_f16x3t_set_mfma(shape)
        try:
            xt, _ = ops.rowstats_t(x, 1e-5)
            ht = ops.linear_t(xt, wt, m, n, k, b, act="gelu", alpha=1.0 / ws)
            got = ops.unpack_act_t(ht, m, n)
            assert float((got.double() - ref).abs().max()) <= 4e-6 * scale, shape
        finally:
            lib.lemon_linear_f16x3t_set_mfma(prev)
    gamma, beta = (1.0 + 0.2 * torch.randn(k, generator=g)).cuda(), (0.1 * torch.randn(k, generator=g)).cuda()
    xs = (x * 3.0 + 0.7).contiguous()
    ln = torch.nn.functional.layer_norm(xs.double(), (k,), gamma.double(), beta.double(), 1e-12)
    ref_ln = _gelu64(ln @ w.double().T + b.double())
    wt2, a2, cs, b2 = ops.fold_layernorm_weight(w, b, gamma, beta, 1.0)
    xt, aff = ops.rowstats_t(xs, 1e-12)
    got = ops.unpack_act_t(ops.linear_t_ln(xt, wt2, m, n, k, b2, act="gelu", alpha=a2, row_aff=aff, colsum=cs), m, n)
    assert float((got.double() - ref_ln).abs().max()) <= 1e-5 * float(ref_ln.abs().max())


def test_token_assembly_without_layernorm(hip):
    from lemon_amd import ops
    g = torch.Generator().manual_seed(1)
    for W in (64, 256, 768, 1280):
        p, cls, pos = torch.randn(3, 16, W, generator=g).cuda(), torch.randn(W, generator=g).cuda(), torch.randn(17, W, generator=g).cuda()
        got = ops.vision_tokens_ln(p, cls, pos, None, None)
        ref = torch.cat([cls.expand(3, 1, W), p], 1) + pos
        assert torch.equal(got, ref), W


def _inputs(ours, n_img, lens, seed=4):
    cfg = ours.cfg
    g = torch.Generator().manual_seed(seed)
    px = torch.randn(n_img, 3, cfg.image_size, cfg.image_size, generator=g)
    return px, caption_ids(cfg, lens, seed)


def _check(got, ref, what, raw=1e-4, normed=1e-5):
    sc = float(ref.abs().max())
    d = float((got - ref).abs().max())
    assert d <= raw * max(1.0, sc), f"{what}: raw max abs diff {d} (scale {sc})"
    dn = float((torch.nn.functional.normalize(got, dim=1) - torch.nn.functional.normalize(ref, dim=1)).abs().max())
    assert dn <= normed, f"{what}: normalised max abs diff {dn}"


@pytest.mark.parametrize("mode", ["f16x3", "f16x3-nofold", "f16x3-lib", "bf16x6", "f32"])
def test_mid_size_towers_vs_hf_in_every_gemm_mode(hip, mode, monkeypatch):
    monkeypatch.setenv("LEMON_GEMM", mode.split("-")[0])
    if mode == "f16x3-nofold":
        monkeypatch.setenv("LEMON_LNFOLD", "0")
    if mode == "f16x3-lib":
        monkeypatch.setenv("LEMON_MLP", "lib")
    vit, bert, ours = hf_pair("mid", seed=1)
    px, ids = _inputs(ours, 5, [2, 40, 9, 17, 9, 3, 31, 9])
    ref_i, ref_t = hf_image_features(vit, ours, px), hf_text_features(bert, ours, ids)
    ours = ours.cuda()
    P = ours.cfg.patch_size
    patches = px.unfold(2, P, P).unfold(3, P, P).permute(0, 2, 3, 1, 4, 5).reshape(5, -1, 3 * P * P).contiguous()
    _check(ours.encode_image(px.cuda()).cpu(), ref_i, mode + "/nchw")
    _check(ours.encode_image(patches.cuda()).cpu(), ref_i, mode + "/patch-major")        # convolution bias through the position rows
    got_t = ours.encode_text(ids.cuda()).cpu()                                            # mixed lengths: grouped inside the tower
    _check(got_t, ref_t, mode + "/text")
    # the same captions one group at a time with the lengths handed over (what pipeline.Embedder does): identical bits
    sel = torch.tensor([2, 4, 7])
    again = ours.encode_text(ids[sel].cuda(), seq_len=9, lengths=torch.tensor([9, 9, 9])).cpu()
    assert torch.equal(again, got_t[sel])


def test_full_size_biomedclip_vs_hf(hip):
    vit, bert, ours = hf_pair("full", seed=2, scale=0.03)
    px, ids = _inputs(ours, 3, [256, 2, 17, 64, 130, 17])
    ref_i, ref_t = hf_image_features(vit, ours, px), hf_text_features(bert, ours, ids)
    ours = ours.cuda()
    _check(ours.encode_image(px.cuda()).cpu(), ref_i, "full/image")
    _check(ours.encode_text(ids.cuda()).cpu(), ref_t, "full/text")
    from lemon_amd import ops
    with torch.no_grad():       # (the default mode ran the folded hand-written chain, not the library form)
        assert ours.text.blocks[0].chain_supported(torch.empty(2, 17, 768, device="cuda")) and ops.gemm_mode() == "f16x3"
    ops.gemm_profiling(True)
    try:
        ours.encode_text(ids[2:3].cuda())
        torch.cuda.synchronize()
        assert ops.gemm_profile_read()["launches"] == 4 * 11 + 1    # QKV, output projection, fc1, fc2 of every layer; the last
                                                                     # layer's three row-wise GEMMs run for the [CLS] rows only (library form)
    finally:
        ops.gemm_profiling(False)


def test_embedder_runs_every_caption_at_its_own_length(hip):
    from lemon_amd.pipeline import Embedder
    vit, bert, ours = hf_pair("mid", seed=3)
    g = torch.Generator().manual_seed(9)
    lens = [int(v) for v in torch.randint(2, 41, (57,), generator=g)]
    ids = caption_ids(ours.cfg, lens, seed=5)
    ref = torch.nn.functional.normalize(hf_text_features(bert, ours, ids), dim=1)
    emb = Embedder(ours, torch.device("cuda"), batch_size=4, text_batch_size=6)
    got = emb.embed_texts(ids)
    emb.raise_if_nonfinite()
    assert emb.text_tokens_run == sum(lens) and emb.fallback_rows == 0 and emb.fold_fallback_rows == 0
    assert float((got.cpu() - ref).abs().max()) <= 1e-5
    # raw uint8 images through the transform + patch-operand path
    u8 = torch.randint(0, 256, (6, 50, 70, 3), dtype=torch.uint8, generator=g)
    from lemon_amd.data import gpu_transform_batch
    px = gpu_transform_batch(u8.cuda(), ours.cfg.image_size).cpu()
    ref_i = torch.nn.functional.normalize(hf_image_features(vit, ours.cpu(), px), dim=1)
    emb = Embedder(ours, torch.device("cuda"), batch_size=4)
    assert float((emb.embed_images(u8).cpu() - ref_i).abs().max()) <= 1e-5


def test_cli_runs_biomed_clip_on_the_synthetic_mimic_stand_in(hip, tmp_path):
    from lemon_amd.run_lemon import main
    out = str(tmp_path / "run")
    rc = main(["--output_dir", out, "--dataset", "mimiccxr_caption", "--noise_type", "random", "--noise_level", "0.3", "--data_root",
               "synthetic:600", "--clip_model", "biomed_clip", "--clip_path", "random:biomed-tiny", "--knn_k", "4", "--hparam_grid", "small"])
    assert rc == 0
    res = pickle.load(open(os.path.join(out, "res.pkl"), "rb"))
    df = res["df"]
    assert set(df.sset.unique()) == {"train", "val", "test"} and df["D_m"].iloc[0].shape == (4,)
    assert np.isfinite(np.stack(list(df["dists_m"]))).all() and 0.0 <= res["agg_results"]["know_val_labels"]["test"]["AUROC"] <= 1.0
