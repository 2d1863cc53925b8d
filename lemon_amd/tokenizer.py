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
