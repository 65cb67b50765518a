"""Golden ids for spn4cir_amd/bert_tokenizer.py.  Build container only (needs `transformers`).

    python tests/golden/make_golden_bert_tokenizer.py   -> bert_tokenizer.json

The reference tokenises BLIP captions with transformers' Python `BertTokenizer` (pinned transformers==4.33.2,
requirements.txt:7) loaded from the hub (blip4cir/blip.py:189-194) and overwrites the first id with [ENC]
(blip_cir.py:87-88).  The hub vocabulary cannot be fetched here, so the capture runs the SAME class and the SAME
call sequence on a synthetic vocab.txt written by this script: transformers 5.x keeps that class as
`BertTokenizerLegacy` (tokenization_bert_legacy.py: BasicTokenizer + WordpieceTokenizer, the 4.33 code) next to the
`tokenizers`-backed `BertTokenizer`; both are run and the rows on which they differ are listed in the fixture (the
legacy class is the reference's; the fast one is recorded for information).

Stored: the vocabulary (list, line order = ids), ~330 fuzzed captions, the unpadded id row of each ([ENC] first, as
after blip_cir.py:88) + the padded length of the whole set, one padded sub-batch verbatim and the padded length of two
more (padding='longest' depends on the batch; the generator asserts right-padding with [PAD] and a 1...10...0 mask
for every row it does not store), [DEC] / [ENC] ids.
"""
import json
import os
import random
import tempfile

OUT = os.path.dirname(os.path.abspath(__file__))

WORDS = """a an the is are was has have with without and or but not no more less much very than of in on at to from by for
dress shirt top tee toptee skirt pants jeans shorts coat jacket sweater blouse sleeve sleeves sleeveless collar neck neckline
v-neck button buttons zipper pocket pockets belt lace floral print printed pattern patterned stripe stripes striped plaid solid
graphic logo text word words letter letters picture image photo dog dogs cat cats bird birds horse people person man woman child
red blue green yellow black white grey gray pink purple orange brown beige navy teal gold silver dark light bright pale
long short longer shorter loose tight fitted flowy sheer shiny matte darker lighter brighter colorful plain fancy casual formal
same different similar color colour shape style length fabric material background foreground left right front back side
shows show showing remove add change make replace put instead facing looking standing sitting running lying two three four one
it its this that these those there here their his her similar only also both other another""".split()
PIECES = ["##s", "##es", "##ed", "##ing", "##er", "##est", "##ly", "##less", "##ness", "##y", "##ish", "##able", "##tion",
          "un", "re", "pre", "over", "under", "multi", "non", "##like", "##wear", "##neck", "##line", "##e", "##d", "##n", "##t"]
EXTRA_CHARS = list("éñüçåøßαβγσς") + list("衣服红色長袖") + ["—", "…", "“", "”", "’", "¿", "·", "。", "、"]


def make_vocab():
    v = ["[PAD]"] + [f"[unused{i}]" for i in range(10)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    ascii_chars = [chr(c) for c in range(33, 127)]
    v += ascii_chars
    v += ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    for ch in EXTRA_CHARS:
        low = ch.lower()
        if low not in v:
            v.append(low)
    v += ["##" + c for c in "衣服σς"]
    for w in WORDS:
        for piece in w.lower().replace("-", " - ").split():
            if piece not in v:
                v.append(piece)
    v += [p for p in PIECES if p not in v]
    v += ["sen", "##ora", "##orita", "cafe", "##s##", "##"]          # accent-stripped targets and odd entries
    seen, out = set(), []
    for t in v:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


def make_captions(rng):
    caps = []
    for _ in range(150):                       # plain FashionIQ / CIRR style
        n = rng.randint(3, 14)
        caps.append(" ".join(rng.choice(WORDS) for _ in range(n)))
    for c in list(caps[:60]):                  # case / punctuation / suffix noise
        toks = c.split()
        for i in range(len(toks)):
            r = rng.random()
            if r < 0.15:
                toks[i] = toks[i].upper()
            elif r < 0.3:
                toks[i] = toks[i].capitalize()
            elif r < 0.4:
                toks[i] += rng.choice(["s", "ed", "ing", "er", "ly", "less", "ish", "xyzq", "tion"])
            elif r < 0.5:
                toks[i] += rng.choice([",", ".", "!!!", "...", "?!", ";", ":", ")", "'s", "\"", "-", "--"])
            elif r < 0.55:
                toks[i] = rng.choice(["(", "[", "#", "@", "$", "##", "##s"]) + toks[i]
        caps.append(rng.choice(["", " ", "  "]).join([" ".join(toks)]) + rng.choice(["", ".", " .", "\n", "\t "]))
    specials = [
        "", " ", "\t\n", "Señora wears a café-coloured DRESS", "señorita señoras cafés", "naïve façade über Ångström",
        "衣服 is red 红色 長袖dress", "dress衣服shirt", "the α and β are σ ς ΑΣ", "İstanbul dress", "dress — longer … “quoted” ’s ¿qué?",
        "has non-breaking spaces　here", "zero​width‍joiner and soft­hyphen", "ctrl\x00char\x07here\x7f end",
        "replacement�char", "line sep para sep", "emoji 😀 is unknown 👗", "x" * 100, "y" * 101, "dress " + "z" * 150 + " shirt",
        "[SEP] in the text", "a [MASK] b[CLS]c [PAD]", "has [ENC] and [DEC] inside", "[sep] lower-case is not special", "[UNK]", "[ SEP ]",
        "##s ## ##ing literal hashes", "sleeve##s", "é combining acute, ñ tilde", "ｆｕｌｌｗｉｄｔｈ ｄｒｅｓｓ", "ǅ titlecase ǆ", "ß sharp s STRASSE",
        "1234 56.78 9,000 3/4 50% #1", "it's isn't don't o'clock rock'n'roll", "e-mail@example.com http://x.y/z?a=b&c=d", "a" * 99 + "s",
        "under_score snake_case camelCase", "tabs\tand\nnewlines\r\nmixed", "   leading and trailing   ", "UPPER lower MiXeD",
        "ﬁne ligature ﬂow", "Ω ohm Å angstrom K kelvin", "한국어 hangul にほんご kana", "dress" + "́" * 3, "́̂", ".", "...", "-",
    ]
    caps += specials
    for _ in range(80):                        # random character soup over the interesting alphabet
        alphabet = list("abcdeinorst ABC  .,!?-'#[]()") + EXTRA_CHARS + ["\t", " ", "​", "😀", "é", "É"]
        caps.append("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 40))))
    return caps


def run(tok, caps):
    enc = tok(caps, padding="longest", return_tensors="pt")          # blip_cir.py:87
    ids = enc.input_ids.clone()
    ids[:, 0] = tok.enc_token_id                                      # blip_cir.py:88
    return ids.tolist(), enc.attention_mask.tolist()


def main():
    from transformers.models.bert.tokenization_bert_legacy import BertTokenizerLegacy
    from transformers import BertTokenizer as FastBert
    rng = random.Random(7)
    vocab = make_vocab()
    caps = make_captions(rng)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "vocab.txt")
        with open(path, "w", encoding="utf-8") as f:
            f.write("\n".join(vocab) + "\n")
        toks = {}
        for name, cls in (("legacy", BertTokenizerLegacy), ("fast", FastBert)):
            tok = cls(path) if name == "legacy" else cls(vocab={t: i for i, t in enumerate(vocab)})
            tok.add_special_tokens({"bos_token": "[DEC]"})                       # blip.py:191
            tok.add_special_tokens({"additional_special_tokens": ["[ENC]"]})     # blip.py:192
            # blip.py:193 reads additional_special_tokens_ids[0]; transformers 5 dropped that property: same id by name
            tok.enc_token_id = tok.convert_tokens_to_ids("[ENC]")
            toks[name] = tok
        leg = toks["legacy"]
        ids, mask = run(leg, caps)
        fids, fmask = run(toks["fast"], caps)
        n = len(caps)
        differ = []
        for i in range(n):
            a = [t for t, m in zip(ids[i], mask[i]) if m]
            b = [t for t, m in zip(fids[i], fmask[i]) if m]
            if a != b:
                differ.append(i)
        subs = [list(range(0, 32)), list(range(150, 214)), sorted(rng.sample(range(n), 48))]
        sub_out = []
        for k, idx in enumerate(subs):
            sids, smask = run(leg, [caps[i] for i in idx])
            entry = {"index": idx, "longest": len(sids[0])}
            if k == 0:
                entry.update(ids=sids, mask=smask)          # one padded block verbatim; the others through their shape
            else:
                assert all([t for t, m in zip(r, mk) if m] == [t for t, m in zip(ids[i], mask[i]) if m]
                           and all(t == leg.pad_token_id for t, m in zip(r, mk) if not m) and sorted(mk, reverse=True) == mk
                           for r, mk, i in zip(sids, smask, idx))
            sub_out.append(entry)
        rows = [[t for t, m in zip(r, mk) if m] for r, mk in zip(ids, mask)]
        assert all(all(t == leg.pad_token_id for t, m in zip(r, mk) if not m) and sorted(mk, reverse=True) == mk
                   for r, mk in zip(ids, mask))
        fixture = {"vocab": vocab, "captions": caps, "rows": rows, "longest": len(ids[0]), "sub_batches": sub_out,
                   "pad_token_id": leg.pad_token_id, "dec_token_id": leg.bos_token_id, "enc_token_id": leg.enc_token_id,
                   "fast_tokenizer_differs_on": differ,
                   "source": "transformers %s BertTokenizerLegacy" % __import__("transformers").__version__}
    with open(os.path.join(OUT, "bert_tokenizer.json"), "w", encoding="utf-8") as f:
        json.dump(fixture, f, ensure_ascii=True)
    unk = leg.unk_token_id
    print(f"{n} captions, vocab {len(vocab)}, longest row {len(ids[0])}, [DEC] {leg.bos_token_id} [ENC] {leg.enc_token_id}, "
          f"rows with [UNK] {sum(unk in r for r in ids)}, fast != legacy on {len(differ)} rows: {[caps[i][:30] for i in differ][:12]}")


if __name__ == "__main__":
    main()
