"""CLIP byte-level BPE tokenizer (clip4cir/clip/clip.py:206-247, clip/simple_tokenizer.py:62-129).

Host-side integer/string work (it is Python in the reference too).  This implementation works on
vocabulary ids throughout: a word becomes the ids of its UTF-8 bytes (the last one in its
end-of-word variant), and the lowest-ranked adjacent pair is merged - all of its occurrences, left
to right - until no adjacent pair is in the merge table.  The table (spn4cir_amd/assets/
clip_bpe_merges.npz) is OpenAI CLIP's merge list converted to id pairs by tools/convert_bpe.py.

Deviation: the reference first runs ftfy.fix_text (mojibake repair, a third-party package that is not
available offline); it is the identity on well-formed text such as the FashionIQ / CIRR captions.
"""
import html
import os

import numpy as np
import torch

try:
    import regex as _re
    _PAT = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       _re.IGNORECASE)
except ImportError:                                   # close approximation with the stdlib engine
    import re as _re
    _PAT = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+|_+",
                       _re.IGNORECASE)

_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "clip_bpe_merges.npz")
SOT, EOT = 49406, 49407


def _byte_order():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    for b in range(256):
        if b not in bs:
            bs.append(b)
    return bs            # position in this list = vocabulary id of the byte


class ClipTokenizer:
    def __init__(self, merges_path=None, context_length=77):
        path = merges_path or os.environ.get("SPN4CIR_BPE_MERGES", _ASSET)
        table = np.load(path)["merges"]
        self.rank = {(int(a), int(b)): r for r, (a, b) in enumerate(table)}
        self.byte_id = {b: i for i, b in enumerate(_byte_order())}
        self.context_length = context_length
        self._cache = {}
        self._text_cache = {}          # whole captions: a training set repeats its captions every epoch
        self._text_cache_max = 1 << 20

    def _bpe(self, word_bytes):
        ids = [self.byte_id[b] for b in word_bytes]
        ids[-1] += 256                                 # end-of-word variant of the last byte
        rank = self.rank
        while len(ids) > 1:
            best, best_r = None, None
            for pair in zip(ids, ids[1:]):
                r = rank.get(pair)
                if r is not None and (best_r is None or r < best_r):
                    best, best_r = pair, r
            if best is None:
                break
            a, b = best
            out, i, n = [], 0, len(ids)
            while i < n:
                if i + 1 < n and ids[i] == a and ids[i + 1] == b:
                    out.append(512 + best_r)
                    i += 2
                else:
                    out.append(ids[i])
                    i += 1
            ids = out
        return ids

    def encode(self, text):
        hit = self._text_cache.get(text)
        if hit is not None:
            return list(hit)                # a fresh list per call: callers may mutate the result
        out = self._encode(text)
        if len(self._text_cache) < self._text_cache_max:
            self._text_cache[text] = tuple(out)
        return out

    def _encode(self, text):
        text = html.unescape(html.unescape(text)).strip()
        text = " ".join(text.split()).lower()
        out = []
        for tok in _PAT.findall(text):
            if tok == "<|startoftext|>":
                out.append(SOT)
                continue
            if tok == "<|endoftext|>":
                out.append(EOT)
                continue
            ids = self._cache.get(tok)
            if ids is None:
                ids = self._cache[tok] = self._bpe(tok.encode("utf-8"))
            out.extend(ids)
        return out

    def __call__(self, texts, context_length=None, truncate=False):
        """clip.tokenize: [SOT] + ids + [EOT], zero padded to context_length; int32 [n, context_length]."""
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        result = np.zeros((len(texts), L), dtype=np.int32)      # filled through numpy: one small torch.tensor per row
        for i, t in enumerate(texts):                           # cost more than the BPE itself
            ids = self.encode(t)
            n = len(ids) + 2
            if n > L:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {L}")
                ids = ids[:L - 2]
                n = L
            row = result[i]
            row[0] = SOT
            row[1:n - 1] = ids
            row[n - 1] = EOT
        return torch.from_numpy(result)


_default = None


def tokenize(texts, context_length=77, truncate=False):
    global _default
    if _default is None:
        _default = ClipTokenizer()
    return _default(texts, context_length, truncate)
