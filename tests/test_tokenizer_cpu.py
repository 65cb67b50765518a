"""The BPE tokenizer against ids captured from the reference's clip.tokenize (tests/golden/tokenizer.npz)."""
import json
import os

import numpy as np
import pytest
import torch


def test_tokenizer_matches_reference(golden_dir):
    from spn4cir_amd.tokenizer import ClipTokenizer, tokenize
    z = np.load(os.path.join(golden_dir, "tokenizer.npz"))
    captions = json.loads(str(z["captions"]))
    ids = tokenize(captions)
    assert ids.dtype == torch.int32 and ids.shape == (len(captions), 77)
    assert np.array_equal(ids.numpy(), z["ids"])
    tok = ClipTokenizer()
    assert tok("is red and has long sleeves")[0, :8].tolist() == [49406, 533, 736, 537, 791, 1538, 19691, 49407]
    with pytest.raises(RuntimeError, match="too long"):
        tok("word " * 100)
    t = tok("word " * 100, truncate=True)
    assert t[0, -1].item() == 49407 and (t[0] != 0).all()
    assert tok("a &amp;amp; b")[0, :5].tolist() == tok("a & b")[0, :5].tolist()      # double html unescape
    # the whole-caption cache: a second call returns the same ids (and the truncated form keeps SOT + 75 tokens + EOT)
    again = tok(captions)
    assert np.array_equal(again.numpy(), z["ids"]) and np.array_equal(tok(captions).numpy(), z["ids"])
    t2 = tok("word " * 100, truncate=True)
    assert torch.equal(t, t2) and t2[0, 0].item() == 49406 and t2.shape == (1, 77)
    assert len(tok._text_cache) >= len(set(captions))
    e = tok.encode("is red and has long sleeves")
    e.append(1)                                                    # a caller mutating its result must not poison the cache
    assert tok.encode("is red and has long sleeves") == e[:-1]


def test_tokenizer_more_captions_match_reference(golden_dir):
    """40 more captions + one truncated one, ids captured from the reference's clip.tokenize
    (tests/golden/make_golden_tokenizer.py): punctuation runs, digits, apostrophes, html entities, non-ASCII, whitespace."""
    import json
    from spn4cir_amd.tokenizer import ClipTokenizer
    z = np.load(os.path.join(golden_dir, "tokenizer_more.npz"))
    captions = json.loads(str(z["captions"]))
    assert len(captions) >= 40
    tok = ClipTokenizer()
    got = tok(captions).numpy()
    bad = [c for c, a, b in zip(captions, got, z["ids"]) if not np.array_equal(a, b)]
    assert not bad, bad
    assert np.array_equal(tok(json.loads(str(z["long_caption"])), truncate=True).numpy(), z["long_ids"])


@pytest.mark.skipif(not os.path.exists("/root/reference/clip4cir/clip/simple_tokenizer.py"), reason="reference absent")
def test_tokenizer_fuzz_against_reference():
    """In the build container only: random caption-like strings through both tokenizers."""
    import importlib.util
    import sys
    import types
    if "ftfy" not in sys.modules:
        m = types.ModuleType("ftfy")
        m.fix_text = lambda s: s
        sys.modules["ftfy"] = m
    spec = importlib.util.spec_from_file_location("_ref_tok", "/root/reference/clip4cir/clip/simple_tokenizer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref = mod.SimpleTokenizer()
    from spn4cir_amd.tokenizer import ClipTokenizer
    mine = ClipTokenizer()
    import random
    rnd = random.Random(0)
    words = ["shorter", "sleeves", "v-neck", "it's", "don't", "darker", "3/4", "floral-print", "café", "naïve", "100%",
             "t-shirt", "(blue)", "asymmetrical", "hemline,", "is", "and", "more", "less", "xxl", "größer", "日本", "a.b",
             "what?!", "#1", "o'clock", "we'll", "they've", "I'm", "x" * 30]
    for _ in range(300):
        s = " ".join(rnd.choice(words) for _ in range(rnd.randint(1, 12)))
        assert mine.encode(s) == ref.encode(s), s
