"""BERT WordPiece tokenizer for the blip4cir fusion encoder (first step).

What the reference runs per step (blip4cir/blip.py:189-194 `init_tokenizer`, blip_cir.py:87-88):

    tokenizer = BertTokenizer.from_pretrained('bert-base-uncased')          # transformers==4.33.2: the Python class
    tokenizer.add_special_tokens({'bos_token': '[DEC]'})                    # id len(vocab)
    tokenizer.add_special_tokens({'additional_special_tokens': ['[ENC]']})  # id len(vocab) + 1
    text = tokenizer(captions, padding='longest', return_tensors='pt')
    text.input_ids[:, 0] = tokenizer.enc_token_id

`BertTokenizer` lives in a third-party package, not under /root/reference; its published algorithm (Devlin et al.'s
tokenization.py as carried by transformers) is restated here from scratch:

  1. text cleaning: drop U+0000, U+FFFD and control characters, map every whitespace character to ' ';
  2. a blank on either side of every CJK ideograph; NFC; split on whitespace;
  3. per word: lower-case, NFD and drop combining marks (category Mn), cut at every punctuation character
     (each punctuation character is its own piece);
  4. per piece: greedy longest-match-first WordPiece with '##' continuation entries; a piece longer than 100
     characters, or one with an unmatched remainder, is one [UNK];
  5. [CLS] ids [SEP]; right-pad with [PAD] to the longest row of the call; attention mask 1 on real tokens.

Literal special tokens inside a caption ('[SEP]', '[MASK]', '[DEC]', '[ENC]', ...) map to their ids unsplit, as the
added-token trie of the reference class does.  Pinned bit-exact against the transformers class on a synthetic
vocabulary (tests/golden/make_golden_bert_tokenizer.py -> bert_tokenizer.json, tests/test_bert_tokenizer_cpu.py); the
real `vocab.txt` of bert-base-uncased is a download the repository cannot vendor: pass its path (`vocab_file=`).

Host-side string work only (as the reference's); whole captions and words are cached, a training set repeats both.
"""
import re
import unicodedata
from typing import Dict, Iterable, List, Sequence, Tuple

import torch

_SPECIALS = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")
_MAX_WORD_CHARS = 100


def _char_class(ch: str) -> int:
    """0 keep, 1 drop, 2 blank, 3 punctuation, 4 CJK ideograph."""
    cp = ord(ch)
    if ch in " \t\n\r":
        return 2
    cat = unicodedata.category(ch)
    if cat == "Zs":
        return 2
    if cp == 0 or cp == 0xFFFD or cat[0] == "C":
        return 1
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126 or cat[0] == "P":
        return 3
    if (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0xF900 <= cp <= 0xFAFF or 0x20000 <= cp <= 0x2A6DF
            or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF
            or 0x2F800 <= cp <= 0x2FA1F):
        return 4
    return 0


def load_vocab(vocab_file: str) -> List[str]:
    """One token per line; the line number is the id (trailing newline stripped, nothing else)."""
    with open(vocab_file, "r", encoding="utf-8") as f:
        return [line.rstrip("\n") for line in f.readlines()]


class Encoding(dict):
    """What `tokenizer(...)` returns: a dict with attribute access and `.to(device)`, enough of BatchEncoding for
    blip_cir.py:87-95 (`text.input_ids`, `text.attention_mask`, `.to(device)`)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def to(self, device):
        return Encoding({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class BertWordPieceTokenizer:
    def __init__(self, vocab_file=None, vocab: Sequence[str] = None, do_lower_case=True, blip_tokens=True):
        if (vocab_file is None) == (vocab is None):
            raise ValueError("give exactly one of vocab_file= (a vocab.txt path) or vocab= (a list of tokens)")
        tokens = load_vocab(vocab_file) if vocab_file is not None else list(vocab)
        self.vocab: Dict[str, int] = {}
        for i, t in enumerate(tokens):
            self.vocab[t] = i                      # a repeated line keeps its last id, as a dict built in file order does
        for s in _SPECIALS:
            if s not in self.vocab:
                raise ValueError(f"vocabulary has no {s} entry")
        self.do_lower_case = do_lower_case
        self.unk_token_id, self.sep_token_id = self.vocab["[UNK]"], self.vocab["[SEP]"]
        self.pad_token_id, self.cls_token_id = self.vocab["[PAD]"], self.vocab["[CLS]"]
        self.mask_token_id = self.vocab["[MASK]"]
        self.added: Dict[str, int] = {}
        self._special = {s: self.vocab[s] for s in _SPECIALS}
        self.bos_token_id = self.enc_token_id = None
        self._max_piece = max((len(t) for t in self.vocab), default=1)
        self._word_cache: Dict[str, Tuple[int, ...]] = {}
        self._caption_cache: Dict[str, Tuple[int, ...]] = {}
        self._cls: Dict[str, int] = {}
        self._rebuild_split()
        if blip_tokens:                            # blip.py:191-193
            self.add_special_tokens({"bos_token": "[DEC]"})
            self.add_special_tokens({"additional_special_tokens": ["[ENC]"]})
            self.enc_token_id = self.additional_special_tokens_ids[0]

    # --------------------------------------------------------------------------------- special tokens
    def __len__(self):
        return len(self.vocab) + len(self.added)

    def _rebuild_split(self):
        names = sorted(self._special, key=len, reverse=True)
        self._split_re = re.compile("(" + "|".join(re.escape(n) for n in names) + ")")
        self._caption_cache.clear()

    def add_special_tokens(self, mapping: dict) -> int:
        """`{'bos_token': str}` and `{'additional_special_tokens': [str, ...]}`: a token that is not in the vocabulary
        takes the next free id (len(vocab), len(vocab) + 1, ...); returns how many were new."""
        new = 0
        self.additional_special_tokens_ids = getattr(self, "additional_special_tokens_ids", [])
        for key, val in mapping.items():
            for tok in ([val] if isinstance(val, str) else list(val)):
                if tok in self.vocab:
                    tid = self.vocab[tok]
                elif tok in self.added:
                    tid = self.added[tok]
                else:
                    tid = len(self.vocab) + len(self.added)
                    self.added[tok] = tid
                    new += 1
                self._special[tok] = tid
                if key == "bos_token":
                    self.bos_token_id = tid
                elif key == "additional_special_tokens":
                    self.additional_special_tokens_ids = self.additional_special_tokens_ids + [tid]
        self._rebuild_split()
        return new

    # --------------------------------------------------------------------------------- words -> ids
    def _wordpiece(self, piece: str, out: List[int]):
        n = len(piece)
        if n > _MAX_WORD_CHARS:
            out.append(self.unk_token_id)
            return
        vocab, ids, start = self.vocab, [], 0
        while start < n:
            end = min(n, start + self._max_piece)
            hit = None
            while end > start:
                sub = piece[start:end] if start == 0 else "##" + piece[start:end]
                hit = vocab.get(sub)
                if hit is not None:
                    break
                end -= 1
            if hit is None:
                out.append(self.unk_token_id)
                return
            ids.append(hit)
            start = end
        out.extend(ids)

    def _word_ids(self, word: str) -> Tuple[int, ...]:
        got = self._word_cache.get(word)
        if got is not None:
            return got
        w = word
        if self.do_lower_case:
            w = w.lower()
            w = "".join(c for c in unicodedata.normalize("NFD", w) if unicodedata.category(c) != "Mn")
        pieces, cur, cls = [], [], self._cls
        for ch in w:
            k = cls.get(ch)
            if k is None:
                k = cls[ch] = _char_class(ch)
            if k == 3:
                if cur:
                    pieces.append("".join(cur))
                    cur = []
                pieces.append(ch)
            else:
                cur.append(ch)
        if cur:
            pieces.append("".join(cur))
        out: List[int] = []
        for p in " ".join(pieces).split():            # a stripped accent can leave an empty or blank piece behind
            self._wordpiece(p, out)
        got = self._word_cache[word] = tuple(out)
        return got

    def _plain_ids(self, text: str, out: List[int]):
        cls, buf = self._cls, []
        for ch in text:
            k = cls.get(ch)
            if k is None:
                k = cls[ch] = _char_class(ch)
            if k == 1:
                continue
            if k == 2:
                buf.append(" ")
            elif k == 4:
                buf.append(" " + ch + " ")
            else:
                buf.append(ch)
        for word in unicodedata.normalize("NFC", "".join(buf)).split():
            out.extend(self._word_ids(word))

    def encode_plain(self, text: str) -> Tuple[int, ...]:
        """ids of one caption without [CLS] / [SEP]."""
        got = self._caption_cache.get(text)
        if got is not None:
            return got
        out: List[int] = []
        for part in self._split_re.split(text):
            if not part:
                continue
            sid = self._special.get(part)
            if sid is not None:
                out.append(sid)
            else:
                self._plain_ids(part, out)
        got = tuple(out)
        if len(self._caption_cache) < 1_000_000:
            self._caption_cache[text] = got
        return got

    def encode(self, text: str) -> List[int]:
        return [self.cls_token_id, *self.encode_plain(text), self.sep_token_id]

    # --------------------------------------------------------------------------------- batches
    def batch(self, texts: Iterable[str]) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (input_ids int64 [B, L], attention_mask int64 [B, L]), L = the longest row (padding='longest')."""
        if isinstance(texts, str):
            texts = [texts]
        rows = [self.encode(t) for t in texts]
        L = max((len(r) for r in rows), default=0)
        if not rows or L == 0:
            return torch.zeros((len(rows), L), dtype=torch.int64), torch.zeros((len(rows), L), dtype=torch.int64)
        # one tensor construction for the whole batch: a per-row slice assignment costs ~15 us of host time each, 2 ms per
        # 128-caption step in front of every device launch of the step (the reference loop syncs on the loss every step)
        pad = self.pad_token_id
        ids = torch.tensor([r + [pad] * (L - len(r)) for r in rows], dtype=torch.int64)
        lens = torch.tensor([len(r) for r in rows], dtype=torch.int64)
        mask = (torch.arange(L, dtype=torch.int64)[None, :] < lens[:, None]).to(torch.int64)
        return ids, mask

    def __call__(self, texts, padding="longest", return_tensors="pt", **unused):
        if padding not in ("longest", True):
            raise ValueError("only padding='longest' (blip_cir.py:87) is implemented")
        if return_tensors != "pt":
            raise ValueError("only return_tensors='pt' is implemented")
        ids, mask = self.batch(texts)
        return Encoding(input_ids=ids, token_type_ids=torch.zeros_like(ids), attention_mask=mask)

    def enc_batch(self, texts) -> Tuple[torch.Tensor, torch.Tensor]:
        """The two lines of blip_cir.py:87-88 in one call: padded ids with the first token replaced by [ENC], mask."""
        ids, mask = self.batch(texts)
        if self.enc_token_id is None:
            raise RuntimeError("no [ENC] token: construct with blip_tokens=True or add it with add_special_tokens")
        if ids.shape[1]:
            ids[:, 0] = self.enc_token_id
        return ids, mask


def init_tokenizer(vocab_file: str) -> BertWordPieceTokenizer:
    """blip4cir/blip.py:189-194 with the vocabulary read from a local vocab.txt instead of the hub."""
    return BertWordPieceTokenizer(vocab_file=vocab_file)
