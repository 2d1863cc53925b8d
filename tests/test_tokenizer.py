"""CPU: lemon_amd/tokenizer.py against ids produced by the reference's SimpleTokenizer + tokenize()
(lib/models/simple_tokenizer.py:86-156, lib/models/chexzero_clip.py:481-493) and by HF CLIPTokenizerFast built offline
from the same merges table (tools/make_golden_tokenizer.py -> tests/golden/tokenizer.npz).  The fixture carries the
sparse (pair, rank) table the texts' merge paths touch, so no vocabulary file is needed here."""
import os

import numpy as np
import pytest
import torch

from lemon_amd.tokenizer import ClipBPE, HFStyleClipTokenizer, find_bpe_file, tokenize

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "tokenizer.npz"))


@pytest.fixture(scope="module")
def bpe():
    ranks = {(str(a), str(b)): int(r) for a, b, r in zip(FX["merge_first"], FX["merge_second"], FX["merge_rank"])}
    return ClipBPE(ranks)


def test_vocabulary_layout(bpe):
    assert bpe.sot_id == 49406 and bpe.eot_id == 49407 and bpe.vocab_size == 49408
    assert bpe.encoder["!"] == 0 and bpe.encoder["!</w>"] == 256


@pytest.mark.parametrize("ctx", [77, 256, 16])
def test_tokenize_matches_reference(bpe, ctx):
    texts = [str(t) for t in FX["texts"]]
    got = tokenize(texts, ctx, bpe)
    assert got.dtype == torch.long and got.shape == (len(texts), ctx)
    assert np.array_equal(got.numpy(), FX[f"ids_ref_{ctx}"])


def test_tokenize_accepts_a_model_object(bpe):
    class M:
        context_length = 77
    assert np.array_equal(tokenize(["A photo of a cat"], M(), bpe).numpy(), FX["ids_ref_77"][28:29])


def test_truncation_forces_eot_last(bpe):
    ids = FX["ids_ref_16"]
    long_rows = [i for i, t in enumerate(FX["texts"]) if len(str(t)) > 200]
    assert long_rows and all(ids[i, 15] == 49407 and (ids[i] != 0).all() for i in long_rows)


def test_hf_style_call_matches_cliptokenizerfast(bpe):
    texts = [str(t) for t in FX["texts"]]
    out = HFStyleClipTokenizer(bpe, 77)(texts, padding="max_length", truncation=True)
    same = FX["hf_same"]
    assert len(same) == len(texts)          # every golden text is clean UTF-8: the two conventions must agree
    assert np.array_equal(np.array(out["input_ids"]), FX["ids_hf"])
    assert np.array_equal(np.array(out["attention_mask"]), FX["mask_hf"])
    # EOT pooling reads the FIRST max id: padding with the EOT id keeps argmax on the real EOT
    plain = [i for i, t in enumerate(texts) if "<|endoftext|>" not in t]
    ids = torch.tensor(out["input_ids"])[plain]
    assert (ids.argmax(-1) == torch.tensor(out["attention_mask"])[plain].sum(-1) - 1).all()


def test_find_bpe_file(tmp_path, monkeypatch):
    monkeypatch.delenv("LEMON_BPE_PATH", raising=False)
    assert find_bpe_file(None) is None
    f = tmp_path / "merges.txt"
    f.write_text("#version: 0.2\ni n\nt h\n")
    assert find_bpe_file(str(tmp_path)) == str(f) and find_bpe_file(str(f)) == str(f)
    monkeypatch.setenv("LEMON_BPE_PATH", str(f))
    assert find_bpe_file(None) == str(f)
    small = ClipBPE.from_file(str(f))
    assert small.ranks == {("i", "n"): 0, ("t", "h"): 1}
