"""Padding vectors of the reference's TargetPad (clip4cir/data_utils.py:42-65), captured by running the class
itself with a recording stand-in for torchvision's F.pad (build container only).

    python tests/golden/make_golden_targetpad.py  ->  tests/golden/targetpad.npz
"""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_golden import REF, install_stubs  # noqa: E402


class _Img:
    def __init__(self, w, h):
        self.size = (w, h)


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "clip4cir"))
    import data_utils  # noqa: E402
    rec = []
    data_utils.F.pad = lambda img, padding, fill, mode: rec.append(tuple(padding)) or img
    rng = np.random.default_rng(7)
    sizes = [(224, 224), (300, 200), (200, 300), (1000, 333), (333, 1000), (640, 480), (500, 400), (401, 500),
             (125, 100), (124, 100), (1, 7), (3000, 200)]
    sizes += [(int(a), int(b)) for a, b in rng.integers(16, 1500, size=(200, 2))]
    rows = []
    for ratio in (1.25, 1.0, 2.0):
        tp = data_utils.TargetPad(ratio, 224)
        for w, h in sizes:
            rec.clear()
            out = tp(_Img(w, h))
            hp, vp = (rec[0][0], rec[0][1]) if rec else (0, 0)
            if rec:
                assert rec[0] == (hp, vp, hp, vp)
            rows.append((ratio, w, h, hp, vp, 1 if rec else 0))
    np.savez_compressed(os.path.join(OUT, "targetpad.npz"), rows=np.array(rows, dtype=np.float64))
    print(len(rows), "cases")


if __name__ == "__main__":
    main()
