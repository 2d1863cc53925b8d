"""The `biomed_clip` branch of the factory (lib/models/utils.py:72-78) on the CPU: towers against HF ViTModel / BertModel on shared
random weights, the open_clip checkpoint loader, the WordPiece tokenizer against transformers' BertTokenizer, the factory surface,
and the exact-length grouping pipeline.Embedder applies to a tower without a padding mask in its kernels."""
import pytest
import torch

from lemon_amd.biomed import BiomedCLIP
from lemon_amd.clip import algorithm_class_from_scratch
from lemon_amd.tokenizer import BertWordPiece, find_vocab_file

from .biomed_recipe import caption_ids, hf_image_features, hf_pair, hf_text_features

pytest.importorskip("transformers")


def test_towers_match_hf_vit_and_bert_on_random_weights():
    vit, bert, ours = hf_pair("tiny")
    g = torch.Generator().manual_seed(3)
    px = torch.randn(5, 3, 32, 32, generator=g)
    ids = caption_ids(ours.cfg, [2, 7, 24, 11, 7, 3])
    ref_i, ref_t = hf_image_features(vit, ours, px), hf_text_features(bert, ours, ids)
    got_i, got_t = ours.encode_image(px), ours.encode_text(ids)
    assert got_i.shape == (5, 32) and got_t.shape == (6, 32)
    assert (got_i - ref_i).abs().max() < 2e-5 * max(1.0, float(ref_i.abs().max())), (got_i - ref_i).abs().max()
    assert (got_t - ref_t).abs().max() < 2e-5 * max(1.0, float(ref_t.abs().max())), (got_t - ref_t).abs().max()
    # a caption's embedding does not depend on its batch mates or on the padded width (what the length grouping relies on)
    alone = ours.encode_text(ids[2:3])
    cut = ours.encode_text(ids[1:2, :7])
    assert (alone - got_t[2:3]).abs().max() < 1e-5 and (cut - got_t[1:2]).abs().max() < 1e-5


def test_padding_inside_a_caption_is_masked_like_hf():
    _, bert, ours = hf_pair("tiny", seed=2)
    ids = caption_ids(ours.cfg, [9, 12])
    ids[0, 4] = 0                                     # a [PAD] in the middle: a masked key, its own row still attends
    ref = hf_text_features(bert, ours, ids)
    assert (ours.encode_text(ids) - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))


def test_open_clip_checkpoint_round_trip(tmp_path):
    _, _, ours = hf_pair("tiny", seed=5)
    sd = ours.open_clip_state_dict()
    assert sd["visual.trunk.cls_token"].shape == (1, 1, 64) and sd["visual.trunk.pos_embed"].shape == (1, 17, 64)
    assert sd["text.proj.0.weight"].shape == (48, 64) and sd["text.proj.2.weight"].shape == (32, 48)
    assert sd["text.transformer.encoder.layer.1.attention.self.key.weight"].shape == (64, 64)
    d = tmp_path / "snap"
    d.mkdir()
    sd["text.transformer.embeddings.position_ids"] = torch.arange(32)[None]        # (a buffer older checkpoints carry)
    torch.save(sd, d / "open_clip_pytorch_model.bin")
    (d / "open_clip_config.json").write_text('{"model_cfg": {"embed_dim": 32, "text_cfg": {"context_length": 24}}}')
    again = BiomedCLIP.from_pretrained(str(d))        # the architecture is read off the tensor shapes
    assert again.cfg == ours.cfg
    ids = caption_ids(ours.cfg, [5, 9])
    assert torch.equal(again.encode_text(ids), ours.encode_text(ids))
    with pytest.raises(FileNotFoundError):
        BiomedCLIP.from_pretrained(str(tmp_path / "nowhere"))
    with pytest.raises(FileNotFoundError):
        algorithm_class_from_scratch("biomed_clip", "hf-hub:microsoft/BiomedCLIP-PubMedBERT_256-vit_base_patch16_224", None)


VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "the", "chest", "x", "ray", "shows", "no", "acute", "card", "##io", "##pul", "##monary",
         "disease", ".", ",", "-", "(", ")", "pleural", "eff", "##usion", "##s", "left", "right", "lung", "is", "clear", "1", "2", "##2", "cm",
         "resume", "nai", "##ve", "a", "of", "##a", "##b", "##c", "b", "c", "d", "##d", "&", ";", "lt", "/", "<", ">", "t", "'", "s", "don"]


def test_wordpiece_matches_transformers_bert_tokenizer(tmp_path):
    from transformers import BertTokenizer
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    hf = BertTokenizer(str(vf), do_lower_case=True)
    ours = BertWordPiece.from_file(str(vf), context_length=16)
    assert find_vocab_file(str(tmp_path)) == str(vf) and find_vocab_file(str(vf)) == str(vf)
    texts = ["The chest X-ray shows no acute cardiopulmonary disease.", "Pleural effusions (left), 12 cm", "Résumé naïve; UNKNOWNWORD lungs",
             "a\tb\n c  d", "lung" * 30, "", "don't   a/b", "the " * 40, "x" + chr(0x4E2D) + "ray", "abc abd ab"]
    enc = hf(texts, padding="max_length", truncation=True, max_length=16, return_tensors="pt").input_ids
    got = ours(texts)
    assert got.shape == (len(texts), 16) and got.dtype == torch.long
    assert torch.equal(got, enc), [(t, a.tolist(), b.tolist()) for t, a, b in zip(texts, got, enc) if not torch.equal(a, b)][:2]
    assert ours(["the lung"], context_length=6).tolist() == [[2, 5, 28, 3, 0, 0]]
    # open_clip's whitespace clean: html entities twice, runs of blanks
    assert ours(["the &amp;lt; lung"]).tolist() == ours(["the < lung"]).tolist()


def test_factory_surface_and_reference_call_convention(tmp_path):
    model, tok = algorithm_class_from_scratch("biomed_clip", "random:biomed-tiny", None, return_tokenizer=True)
    assert isinstance(model, BiomedCLIP) and model.context_length == 24
    toks = tok(["a chest x ray", "no acute disease in the lungs"])            # run_lemon.py:148-149: tokenizer(texts) -> tensor
    assert toks.shape == (2, 24) and toks.dtype == torch.long and toks[0, 0] == 2 and toks[1, 7] == 3 and toks[1, 8] == 0
    emb = model.encode_text(toks)                                              # :157-158
    assert emb.shape == (2, 32) and torch.isfinite(emb).all()
    # with a vocabulary file the WordPiece tokenizer is used
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    _, tok2 = algorithm_class_from_scratch("biomed_clip", "random:biomed-tiny", None, return_tokenizer=True, bpe_path=str(vf))
    assert tok2(["the lung is clear"]).tolist()[0][:6] == [2, 5, 28, 29, 30, 3]
    assert algorithm_class_from_scratch("biomed_clip", "random", None).cfg.context_length == 256


def test_embedder_groups_captions_by_exact_length(monkeypatch):
    """pipeline.Embedder on a tower with `exact_lengths`: captions sorted by token count, micro-batches never mix counts, the
    per-row counts travel to the tower, results come back in caption order."""
    from lemon_amd.pipeline import Embedder
    _, _, ours = hf_pair("tiny", seed=7)
    emb = Embedder.__new__(Embedder)
    emb.model, emb.device, emb.text_batch_size, emb.length_bucketing, emb.range_fallback, emb.text_tokens_run = ours, torch.device("cpu"), 3, False, False, 0
    emb.text_token_budget = None
    lens = [9, 4, 9, 2, 9, 9, 4, 24, 9]
    ids = caption_ids(ours.cfg, lens)
    seen = []
    real = ours.encode_text

    def spy(rows, seq_len=None, lengths=None, **kw):
        seen.append((int(seq_len), lengths.tolist()))
        return real(rows)

    monkeypatch.setattr(ours, "encode_text", spy)
    out = emb._embed_texts(ids)
    assert [s for s, _ in seen] == [2, 4, 9, 9, 24] and all(set(l) == {s} for s, l in seen) and [len(l) for _, l in seen] == [1, 2, 3, 2, 1]
    assert emb.text_tokens_run == sum(lens)
    assert (out - real(ids)).abs().max() < 1e-5
