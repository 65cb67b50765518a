"""TIMING PROBE ONLY (the overlapped run races AdamW against the next forward: its numbers are garbage): what would the step gain
if the optimizer update ran on a side stream under the next step's forward pass?   python tools/overlap_probe.py [packed|dense]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import ops, synthetic
from spn4cir_amd.models import CIRPlus
from spn4cir_amd.trainer import Stage2Trainer
mode = sys.argv[1] if len(sys.argv) > 1 else "packed"
dev = torch.device("cuda")
W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
model = CIRPlus(synthetic.text_state_dict(W, layers, D, seed=0), tau=0.02, device=dev, plus=True)
target, refer = synthetic.banks(40000, D, seed=2)
tr = Stage2Trainer(model, lr=2e-5, pack=False)
tr.set_banks(refer, target)
B = 256
ids_h = synthetic.token_ids(B, seed=1)
ridx, lab = synthetic.triplet_indices(B, 40000, seed=4)
ids, ridx, lab = ids_h.to(dev), ridx.to(dev), lab.to(dev)
cu, total = (None, 0)
if mode == "packed":
    cu, total = tr.tower.cu_seqlens(ids_h)
    cu = cu.to(dev)
    ids = ids[:, :tr.tower.live_length(ids_h)].contiguous()

def run(n):
    for _ in range(3):
        tr.step(ids, ridx, lab, cu, total)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.step(ids, ridx, lab, cu, total)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

base = run(20)
side = torch.cuda.Stream()
orig = ops.adamw_step
def adamw_side(*a, **k):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        return orig(*a, **k)
ops.adamw_step = adamw_side
over = run(20)
print(f"{mode}: serial {base:.3f} ms/step, AdamW on a side stream under the next forward {over:.3f} ms/step")
