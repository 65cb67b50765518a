"""More tokenizer vectors (SURVEY asked for ~50 captions; make_golden.py holds the first 15): ids captured from the
reference's clip.tokenize (clip4cir/clip/clip.py:206-247, simple_tokenizer.py:62-134) in the build container.

    python tests/golden/make_golden_tokenizer.py   ->  tokenizer_more.npz (captions as a json string, int32 ids [n, 77])

FashionIQ / CIRR style relative captions, the 4-way FashionIQ joins, punctuation runs, digits, apostrophes, html entities
(double unescape), non-ASCII letters, whitespace runs, upper case, and one caption cut by truncate=True."""
import json
import os
import sys
import types

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
CAPTIONS = [
    "is more fitted and has a zipper", "has spaghetti straps and a sweetheart neckline", "is strapless, shorter and pink",
    "has a crew neck and three-quarter sleeves", "is a tank top with stripes and is brighter in color",
    "the shirt is lighter and has a pocket on the left side", "is blue with white polka dots and cap sleeves",
    "has longer sleeves and is darker and is more revealing with a lower neckline",
    "is more see-through and lacy and has a high-low hem", "less busy pattern, more solid in colour",
    "shows the same dog breed lying on a couch", "zoom in on the bird and make the background blurry",
    "has two of them and they are on a wooden table", "remove the child, add a bicycle next to the bench",
    "the monkey faces the camera and there is no fence", "is at night with street lights on", "same bottle but it's empty and tipped over",
    "instead of a laptop, a tablet is held by a woman", "change the plate to a bowl & add chopsticks",
    "a t-shirt that says 'I love NY' in red", "it's 50% shorter...", "what?! no sleeves??", "size XL, not XXL",
    "costs $20 (was $35)", "#1 best-seller: off-the-shoulder", "café au lait coloured, naïve print",
    "größer und dunkler", "日本の着物 style with an obi", "a &amp; b", "a &amp;amp; b &lt;3",
    "   leading   and   trailing   spaces   ", "ALL CAPS AND Mixed Case", "tabs\tand\nnewlines become spaces",
    "hyphen-ated-words and under_scores", "e.g. i.e. etc.", "1 2 3 4 5 6 7 8 9 10", "3/4 sleeves w/ a 1.5\" belt",
    "don't won't can't they're we've I'm o'clock", "is red and has long sleeves and is shorter and more colorful",
    "is shorter and more colorful and is red and has long sleeves",
]
LONG = "very " * 120 + "long"


def main():
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s
    sys.modules["ftfy"] = ftfy
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")

    class _T:
        def __init__(self, *a, **k): pass
        def __call__(self, x): return x
    for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize"):
        setattr(tvt, n, _T)
    tvt.InterpolationMode = type("IM", (), {"BICUBIC": 3})
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})
    sys.path.insert(0, "/root/reference/clip4cir")
    import clip
    ids = clip.tokenize(CAPTIONS).numpy().astype(np.int32)
    cut = clip.tokenize([LONG], truncate=True).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "tokenizer_more.npz"), captions=json.dumps(CAPTIONS), ids=ids,
                        long_caption=json.dumps(LONG), long_ids=cut)
    print("wrote tokenizer_more.npz", ids.shape, cut.shape)


if __name__ == "__main__":
    main()
