"""CPU oracle for the SPN4CIR stage-2 hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (PyTorch fp32 / numpy) restatement of the
reference algorithm for the path named in BASELINE.json's north_star.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the CPU baseline, never as the product path.
The product (``spn4cir_amd``) never imports ``oracle`` and raises when its HIP library is
missing.

Pinning: every function here is checked in ``tests/test_oracle_golden.py`` against golden
vectors captured by importing the reference implementation itself
(``tests/golden/make_golden.py``, run in the build container where /root/reference exists).
The reference ships no tests of its own, so those captured vectors
are the only pin available.

Each function cites the reference file:line it restates (paths relative to the reference
repository root).
"""
from . import clip_text, clip_vision, bank_loss, recall, optim, bert_fusion  # noqa: F401
