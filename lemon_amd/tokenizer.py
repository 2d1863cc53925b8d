"""Byte-pair-encoding tokenizer of the CLIP family, for the tokenizer call of the hot path
(run_lemon.py:149,151-154,183,186-189,220,222-226).

Two call conventions the reference uses, both served from ONE merges table:
  * `tokenize(texts, model)`            lib/models/chexzero_clip.py:481-493 (the in-tree CLIP branches of
    lib/models/utils.py:82-103): LongTensor [n, model.context_length], SOT + BPE ids + EOT, zero padded, over-long
    prompts cut to the context length with EOT forced into the last slot;
  * `tokenizer(list[str], padding="max_length", truncation=True)`  the HF CLIPTokenizer(Fast) call of the
    'huggingface_clip' branch: dict of lists, padded with the EOT id (openai/clip-vit-* use <|endoftext|> as pad),
    truncated to 77 with EOT last.
The BPE itself follows the published CLIP algorithm (OpenAI CLIP `simple_tokenizer.py`, which
lib/models/simple_tokenizer.py:86-156 vendors): lower-case, whitespace-collapse, split with the CLIP pattern,
bytes -> printable code points, then merge the lowest-ranked adjacent pair until none is ranked.

The merges table is DATA the user supplies (`bpe_simple_vocab_16e6.txt.gz`, or HF's `merges.txt`): nothing is
downloaded and no vocabulary ships with this package.  Text cleaning: `ftfy.fix_text` when ftfy is importable
(the reference requires it, simple_tokenizer.py:30,74-77), otherwise skipped -- identical on clean UTF-8 text.
"""
import gzip
import html
import os
from functools import lru_cache

import regex as re

SOT, EOT = "<|startoftext|>", "<|endoftext|>"
N_MERGES = 49152 - 256 - 2            # merges kept from the 16e6 vocabulary file (simple_tokenizer.py:91)
_PATTERN = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""


@lru_cache()
def byte_alphabet():
    """byte -> printable unicode character (the 256 base symbols), in vocabulary order: the 188 printable latin-1
    bytes keep their own code point, the 68 others are mapped to 256, 257, ..."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    order, chars, extra = list(keep), [chr(b) for b in keep], 0
    for b in range(256):
        if b not in keep:
            order.append(b)
            chars.append(chr(256 + extra))
            extra += 1
    return dict(zip(order, chars))


def read_merges(path):
    """Merge rules in rank order from `bpe_simple_vocab_16e6.txt(.gz)` (header line + 'a b' lines; the first
    N_MERGES are used) or from an HF `merges.txt` (same format, already cut)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        lines = f.read().decode("utf-8").split("\n")
    lines = lines[1:N_MERGES + 1]
    return [tuple(l.split()) for l in lines if len(l.split()) == 2]


class ClipBPE:
    """encode(text) -> list of ids.  `merges`: [(first, second), ...] in rank order, or {(first, second): rank}
    (a sparse table is enough to tokenise texts whose merge paths it covers: the tests' subset fixture)."""

    def __init__(self, merges):
        ranks = dict(merges) if isinstance(merges, dict) else {tuple(m): i for i, m in enumerate(merges)}
        self.ranks = ranks
        alphabet = list(byte_alphabet().values())
        self.encoder = {c: i for i, c in enumerate(alphabet)}
        self.encoder.update({c + "</w>": 256 + i for i, c in enumerate(alphabet)})
        for (a, b), r in sorted(ranks.items(), key=lambda kv: kv[1]):
            self.encoder[a + b] = 512 + r
        self.sot_id = 512 + N_MERGES
        self.eot_id = self.sot_id + 1
        self.encoder[SOT], self.encoder[EOT] = self.sot_id, self.eot_id
        self.vocab_size = self.eot_id + 1
        self._cache = {SOT: (SOT,), EOT: (EOT,)}
        self._pat = re.compile(_PATTERN, re.IGNORECASE)
        self._bytes = byte_alphabet()

    @classmethod
    def from_file(cls, path):
        return cls(read_merges(path))

    def _bpe(self, token):
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word[:-1], word[1:]):
                r = self.ranks.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    merged.append(best[0] + best[1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = tuple(word)
        self._cache[token] = out
        return out

    @staticmethod
    def clean(text):
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return re.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text):
        ids = []
        for tok in re.findall(self._pat, self.clean(text)):
            tok = "".join(self._bytes[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(tok))
        return ids


def tokenize(texts, model_or_context_length, bpe):
    """lib/models/chexzero_clip.py:481-493: LongTensor [n, context_length], zero padded."""
    import torch
    ctx = model_or_context_length if isinstance(model_or_context_length, int) else model_or_context_length.context_length
    out = torch.zeros(len(texts), ctx, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [bpe.sot_id] + bpe.encode(t) + [bpe.eot_id]
        if len(ids) > ctx:
            ids = ids[:ctx]
            ids[ctx - 1] = bpe.eot_id
        out[i, :len(ids)] = torch.tensor(ids)
    return out


class HFStyleClipTokenizer:
    """The call the 'huggingface_clip' branch makes (run_lemon.py:151-154): tokenizer(texts, padding="max_length",
    truncation=True) -> {"input_ids": [[...]], "attention_mask": [[...]]}, padded to 77 with the EOT id like
    openai/clip-vit-* tokenizers, built from a local merges file when no HF tokenizer directory is available."""

    def __init__(self, bpe, context_length=77):
        self.bpe, self.model_max_length = bpe, context_length

    def __call__(self, texts, padding="max_length", truncation=True, **_):
        ctx = self.model_max_length
        ids_all, mask_all = [], []
        for t in texts:
            ids = [self.bpe.sot_id] + self.bpe.encode(t) + [self.bpe.eot_id]
            if truncation and len(ids) > ctx:
                ids = ids[:ctx - 1] + [self.bpe.eot_id]
            n = len(ids)
            if padding == "max_length":
                ids = ids + [self.bpe.eot_id] * (ctx - n)
            ids_all.append(ids)
            mask_all.append([1] * n + [0] * (len(ids) - n))
        return {"input_ids": ids_all, "attention_mask": mask_all}


def find_bpe_file(hint=None):
    """Merges file from, in order: the explicit hint (file, or a directory holding merges.txt /
    bpe_simple_vocab_16e6.txt.gz), $LEMON_BPE_PATH.  None when nothing is found."""
    for cand in (hint, os.environ.get("LEMON_BPE_PATH")):
        if not cand:
            continue
        if os.path.isdir(cand):
            for fn in ("bpe_simple_vocab_16e6.txt.gz", "merges.txt"):
                if os.path.exists(os.path.join(cand, fn)):
                    return os.path.join(cand, fn)
        elif os.path.exists(cand):
            return cand
    return None


# ---- BERT WordPiece: the tokenizer of the `biomed_clip` branch (lib/models/utils.py:74; run_lemon.py:148-149) ------------------
# open_clip's `get_tokenizer('hf-hub:microsoft/BiomedCLIP-...')` is an HFTokenizer around the checkpoint's BERT uncased
# tokenizer, called as `tokenizer(texts)` -> LongTensor [n, 256]: whitespace-cleaned text, [CLS] pieces [SEP], cut to the
# context length ([SEP] kept last), padded with [PAD].  The algorithm is the published BERT one (BasicTokenizer: clean, CJK
# spacing, lower-case + accent stripping, punctuation split; then greedy longest-match-first WordPiece with '##' continuations),
# restated here from its description; tests/test_biomed.py compares it with `transformers.BertTokenizer`, token for token.  The vocabulary (vocab.txt of the checkpoint) is DATA the user supplies.
import unicodedata


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class BertWordPiece:
    """tokenizer(texts, context_length=None) -> LongTensor [n, context_length] (open_clip HFTokenizer.__call__)."""

    def __init__(self, vocab, context_length=256, lower_case=True, max_chars_per_word=100):
        self.vocab = dict(vocab) if isinstance(vocab, dict) else {t: i for i, t in enumerate(vocab)}
        self.context_length, self.lower_case, self.max_chars = context_length, lower_case, max_chars_per_word
        for t in ("[PAD]", "[UNK]", "[CLS]", "[SEP]"):
            if t not in self.vocab:
                raise ValueError(f"vocabulary lacks {t}")
        self.pad, self.unk, self.cls, self.sep = (self.vocab[t] for t in ("[PAD]", "[UNK]", "[CLS]", "[SEP]"))
        self.never_split = {"[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"}

    @classmethod
    def from_file(cls, path, context_length=256, lower_case=True):
        with open(path, encoding="utf-8") as f:
            toks = [l.rstrip("\n") for l in f]
        while toks and toks[-1] == "":
            toks.pop()
        return cls(toks, context_length, lower_case)

    # -- BasicTokenizer --
    def _clean(self, text):
        out = []
        for ch in text:
            cp = ord(ch)
            cat = unicodedata.category(ch)
            if cp == 0 or cp == 0xFFFD or (cat.startswith("C") and ch not in "\t\n\r"):
                continue
            out.append(" " if ch in " \t\n\r" or cat == "Zs" else ch)
        return "".join(out)

    def _split_word(self, w):
        if self.lower_case and w not in self.never_split:
            w = "".join(c for c in unicodedata.normalize("NFD", w.lower()) if unicodedata.category(c) != "Mn")
        if w in self.never_split:
            return [w]
        parts, cur = [], ""
        for ch in w:
            if _is_punct(ch):
                if cur:
                    parts.append(cur)
                parts.append(ch)
                cur = ""
            else:
                cur += ch
        if cur:
            parts.append(cur)
        return parts

    def basic(self, text):
        text = self._clean(text)
        text = "".join(f" {c} " if _is_cjk(ord(c)) else c for c in text)
        text = unicodedata.normalize("NFC", text)
        return [p for w in text.split() for p in self._split_word(w)]

    def wordpiece(self, word):
        if len(word) > self.max_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, hit = len(word), None
            while start < end:
                piece = ("##" if start else "") + word[start:end]
                if piece in self.vocab:
                    hit = self.vocab[piece]
                    break
                end -= 1
            if hit is None:
                return [self.unk]
            ids.append(hit)
            start = end
        return ids

    def encode(self, text):
        """ids without the special tokens"""
        return [i for w in self.basic(text) for i in self.wordpiece(w)]

    @staticmethod
    def clean_text(text):
        # open_clip's default `clean='whitespace'`: basic_clean (ftfy.fix_text when importable, html.unescape twice) + whitespace
        # collapse -- ftfy skipped when absent, as for the CLIP BPE above
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split())

    def __call__(self, texts, context_length=None):
        import torch
        if isinstance(texts, str):
            texts = [texts]
        ctx = context_length or self.context_length
        out = torch.full((len(texts), ctx), self.pad, dtype=torch.long)
        for r, t in enumerate(texts):
            ids = [self.cls] + self.encode(self.clean_text(str(t)))[:ctx - 2] + [self.sep]
            out[r, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out


def find_vocab_file(hint=None):
    """vocab.txt of a BERT tokenizer: `hint` (file, or a directory holding it), else $LEMON_VOCAB_PATH."""
    for h in (hint, os.environ.get("LEMON_VOCAB_PATH")):
        if not h:
            continue
        if os.path.isfile(h) and not h.endswith((".bin", ".pt", ".safetensors")):
            return h
        d = h if os.path.isdir(h) else os.path.dirname(h)
        if d and os.path.isfile(os.path.join(d, "vocab.txt")):
            return os.path.join(d, "vocab.txt")
    return None
