"""Seeded input builders shared by make_golden.py (capture) and the tests (replay).

Inputs that are too big to commit (the 4 099x512 and 40 000x768 banks) are regenerated
from the torch CPU generator, which is bit-identical across machines; only the reference's
OUTPUTS for them are stored in loss_cases.npz."""
import torch

LOSS_CASES = [(4, 500, 64, 0.01), (32, 4099, 512, 0.02), (16, 40000, 768, 0.03)]


def loss_case_inputs(ci):
    b, m, d, tau = LOSS_CASES[ci]
    g = torch.Generator().manual_seed(100 + ci)
    text = torch.randn(b, d, generator=g)
    refer_bank = torch.randn(m, d, generator=g)
    bank = torch.nn.functional.normalize(torch.randn(m, d, generator=g))
    ridx = torch.randint(0, m, (b,), generator=g)
    labels = torch.randint(0, m, (b,), generator=g)
    if ci == 1:   # label at the first row, the last row, and a duplicate label
        labels[0], labels[1], labels[2] = 0, m - 1, labels[3]
    return text, refer_bank, bank, ridx, labels, tau
