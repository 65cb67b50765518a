"""spn4cir_amd.bert_tokenizer against ids captured from transformers' Python BertTokenizer (the class the reference
loads at blip4cir/blip.py:189-194) on a synthetic vocabulary: bit-exact, incl. padding='longest' and the [ENC] overwrite
of blip_cir.py:87-88.  Fixture: tests/golden/make_golden_bert_tokenizer.py."""
import json
import os

import pytest
import torch

from spn4cir_amd.bert_tokenizer import BertWordPieceTokenizer, init_tokenizer


@pytest.fixture(scope="module")
def fx(golden_dir):
    with open(os.path.join(golden_dir, "bert_tokenizer.json"), encoding="utf-8") as f:
        return json.load(f)


def _padded(rows, L, pad):
    return [r + [pad] * (L - len(r)) for r in rows], [[1] * len(r) + [0] * (L - len(r)) for r in rows]


def test_rows_bit_exact(fx):
    tok = BertWordPieceTokenizer(vocab=fx["vocab"])
    assert tok.bos_token_id == fx["dec_token_id"] and tok.enc_token_id == fx["enc_token_id"]
    assert len(tok) == len(fx["vocab"]) + 2
    bad = []
    for cap, want in zip(fx["captions"], fx["rows"]):
        got = tok.encode(cap)
        got[0] = tok.enc_token_id
        if got != want:
            bad.append((cap, got, want))
    assert not bad, bad[:3]
    assert len(fx["captions"]) >= 200 and sum(tok.unk_token_id in r for r in fx["rows"]) > 20


def test_whole_set_padding_and_enc(fx):
    tok = BertWordPieceTokenizer(vocab=fx["vocab"])
    ids, mask = tok.enc_batch(fx["captions"])
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (len(fx["captions"]), fx["longest"])
    want_ids, want_mask = _padded(fx["rows"], fx["longest"], fx["pad_token_id"])
    assert ids.tolist() == want_ids and mask.tolist() == want_mask


def test_sub_batches_pad_to_their_own_longest(fx):
    tok = BertWordPieceTokenizer(vocab=fx["vocab"])
    for sub in fx["sub_batches"]:
        caps = [fx["captions"][i] for i in sub["index"]]
        enc = tok(caps, padding="longest", return_tensors="pt")          # the reference's call shape
        ids = enc.input_ids.clone()
        ids[:, 0] = tok.enc_token_id
        assert ids.shape[1] == sub["longest"]
        want_ids, want_mask = _padded([fx["rows"][i] for i in sub["index"]], sub["longest"], fx["pad_token_id"])
        assert ids.tolist() == want_ids and enc.attention_mask.tolist() == want_mask
        if "ids" in sub:
            assert ids.tolist() == sub["ids"] and enc["attention_mask"].tolist() == sub["mask"]
        assert enc.to("cpu").input_ids.shape == ids.shape


def test_vocab_file_and_cache(fx, tmp_path):
    path = tmp_path / "vocab.txt"
    path.write_text("\n".join(fx["vocab"]) + "\n", encoding="utf-8")
    tok = init_tokenizer(str(path))
    first = [tok.encode(c) for c in fx["captions"][:50]]
    again = [tok.encode(c) for c in fx["captions"][:50]]               # served from the caption cache
    assert first == again
    for got, want in zip(first, fx["rows"][:50]):
        assert got[1:] == want[1:] and got[0] == tok.cls_token_id
    with pytest.raises(ValueError):
        BertWordPieceTokenizer(vocab=["a", "b"])                        # no [UNK] / [CLS] / ...
    with pytest.raises(ValueError):
        BertWordPieceTokenizer()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only (live transformers class)")
def test_live_fuzz_against_transformers(fx):
    """Fresh random captions against the live class (nothing stored): guards the fixture against over-fitting."""
    import random
    legacy = pytest.importorskip("transformers.models.bert.tokenization_bert_legacy")
    path = "/tmp/_spn_bert_vocab.txt"
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(fx["vocab"]) + "\n")
    ref = legacy.BertTokenizerLegacy(path)
    ref.add_special_tokens({"bos_token": "[DEC]"})
    ref.add_special_tokens({"additional_special_tokens": ["[ENC]"]})
    tok = BertWordPieceTokenizer(vocab=fx["vocab"])
    rng = random.Random(99)
    alphabet = list("abcdefghilnorstuy    .,!?-'#()[]") + list("éÉñßσΣ衣服—…") + ["\t", " ", "​", "😀", "[SEP]", "[MASK]", "##"]
    caps = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 60))) for _ in range(400)]
    want = ref(caps, padding="longest", return_tensors="pt")
    ids, mask = tok.batch(caps)
    assert ids.tolist() == want.input_ids.tolist() and mask.tolist() == want.attention_mask.tolist()
