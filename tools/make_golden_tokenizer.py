"""Golden vectors for lemon_amd/tokenizer.py, generated in the build container from
  * the reference's own SimpleTokenizer + tokenize()  (lib/models/simple_tokenizer.py, lib/models/chexzero_clip.py:481-493;
    ftfy is absent from this image and stubbed with the identity -- exact for the clean UTF-8 texts used here), and
  * HF transformers' CLIPTokenizerFast built OFFLINE from the same merges table (the tokenizer class the
    'huggingface_clip' branch loads by name, lib/models/utils.py:66).
Stores texts, the expected ids, and the SPARSE merges table (pair, rank) those texts' merge paths touch, so the tests can
rebuild a tokenizer that reproduces the ids without shipping the 1.3 MB vocabulary.  Also asserts, here, that the
full-vocabulary ClipBPE agrees with the reference on a larger corpus (all label sets + synthetic captions).
Run:  python tools/make_golden_tokenizer.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

ftfy = types.ModuleType("ftfy")
ftfy.fix_text = lambda t: t
sys.modules["ftfy"] = ftfy
from lib.models import simple_tokenizer as st          # noqa: E402
from lib.models.chexzero_clip import tokenize as ref_tokenize      # noqa: E402
from lemon_amd import tokenizer as T                    # noqa: E402

BPE_FILE = st.default_bpe()
ref_tok = st.SimpleTokenizer()

labels = json.load(open(os.path.join(ROOT, "lemon_amd", "data", "label_sets.json")))["labels"]
TEXTS = ["A photo of a " + l for l in labels["cifar100"][:25]] + ["A photo of a " + l for l in labels["cifar10"]] + [
    "", "a", "A man riding a wave on top of a surfboard.", "Two dogs' owners aren't here; they've left at 10:45pm!!",
    "  multiple   spaces\tand\nnewlines  ", "café naïve façade — résumé", "emoji \U0001F600 test ❤",
    "numbers 1234567890 and 3.14159", "UPPER lower MiXeD", "it's I'm you're we've he'll she'd don't",
    "No acute cardiopulmonary process. Heart size is normal; lungs are clear without focal consolidation.",
    "<|startoftext|> literal special <|endoftext|>", "hyphen-ated under_score slash/and\\back @#$%^&*()",
    " ".join(f"word{i}" for i in range(120)),                       # > 77 tokens: truncation with EOT forced last
    " ".join(["supercalifragilisticexpialidocious"] * 30),
]

# ---- reference ids (in-tree CLIP branches)
Ctx = lambda n: types.SimpleNamespace(context_length=n)
ids_ref = {n: ref_tokenize(TEXTS, Ctx(n)).numpy() for n in (77, 256, 16)}

# ---- HF CLIPTokenizerFast built offline from the same table
from transformers import CLIPTokenizerFast      # noqa: E402
merge_list = sorted(ref_tok.bpe_ranks, key=ref_tok.bpe_ranks.get)
hf = CLIPTokenizerFast(vocab=dict(ref_tok.encoder), merges=[tuple(m) for m in merge_list], model_max_length=77)
enc = hf(TEXTS, padding="max_length", truncation=True)
ids_hf, mask_hf = np.array(enc["input_ids"], np.int64), np.array(enc["attention_mask"], np.int64)

# ---- full-vocabulary agreement on a larger corpus (checked here, where the vocabulary exists)
bpe = T.ClipBPE.from_file(BPE_FILE)
assert bpe.vocab_size == len(ref_tok.encoder) == 49408 and bpe.eot_id == ref_tok.encoder["<|endoftext|>"]
corpus = TEXTS + ["A photo of a " + l for k in labels for l in labels[k]] + \
    [f"a synthetic caption number {i} about category {i % 12}" for i in range(300)]
for n in (77, 256):
    assert torch.equal(T.tokenize(corpus, n, bpe), ref_tokenize(corpus, Ctx(n))), "ClipBPE differs from the reference tokenizer"
mine_hf = T.HFStyleClipTokenizer(bpe, 77)(TEXTS, padding="max_length", truncation=True)
hf_same = [i for i in range(len(TEXTS)) if mine_hf["input_ids"][i] == enc["input_ids"][i] and mine_hf["attention_mask"][i] == enc["attention_mask"][i]]
hf_diff = sorted(set(range(len(TEXTS))) - set(hf_same))
print("HF-style call identical on", len(hf_same), "of", len(TEXTS), "texts; differing:", [TEXTS[i][:40] for i in hf_diff])

# ---- sparse merges table: every ranked pair the merge paths of TEXTS query
touched = {}
orig_get = bpe.ranks.get


class Spy(dict):
    def get(self, k, d=None):
        r = dict.get(self, k, d)
        if r is not None:
            touched[k] = r
        return r


bpe2 = T.ClipBPE(Spy(bpe.ranks))
bpe2.ranks = Spy(bpe.ranks)
T.tokenize(TEXTS, 256, bpe2)
pairs = sorted(touched.items(), key=lambda kv: kv[1])
sparse = T.ClipBPE({p: r for p, r in pairs})
assert torch.equal(T.tokenize(TEXTS, 77, sparse), torch.from_numpy(ids_ref[77]))

np.savez_compressed(os.path.join(OUT, "tokenizer.npz"), texts=np.array(TEXTS), ids_ref_77=ids_ref[77], ids_ref_256=ids_ref[256],
                    ids_ref_16=ids_ref[16], ids_hf=ids_hf, mask_hf=mask_hf, hf_same=np.array(hf_same, np.int64),
                    merge_first=np.array([p[0] for p, _ in pairs]), merge_second=np.array([p[1] for p, _ in pairs]),
                    merge_rank=np.array([r for _, r in pairs], np.int64))
print("tokenizer.npz:", len(TEXTS), "texts,", len(pairs), "merge rules,", os.path.getsize(os.path.join(OUT, "tokenizer.npz")), "B")
