"""Packed-mode stage-2 steps only (for profiling): config 2 with the text tower on the live rows.
    python tools/packed_step.py [steps] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import synthetic
from spn4cir_amd.models import CIRPlus
from spn4cir_amd.trainer import Stage2Trainer
dev = torch.device("cuda")
W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
model = CIRPlus(synthetic.text_state_dict(W, layers, D, seed=0), tau=0.02, device=dev, plus=True)
target, refer = synthetic.banks(40000, D, seed=2)
tr = Stage2Trainer(model, lr=2e-5, pack=False)      # this tool passes cu_seqlens itself
tr.set_banks(refer, target)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256          # triplets per step (second argument)
ids_h = synthetic.token_ids(B, seed=1)
ridx, lab = synthetic.triplet_indices(B, 40000, seed=4)
ids, ridx, lab = ids_h.to(dev), ridx.to(dev), lab.to(dev)
cu, total = tr.tower.cu_seqlens(ids_h)
cu = cu.to(dev)
ids = ids[:, :tr.tower.live_length(ids_h)].contiguous()          # padding columns only beyond the longest caption
for _ in range(3):
    tr.step(ids, ridx, lab, cu, total)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n):
    tr.step(ids, ridx, lab, cu, total)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"packed: B = {B}: {dt * 1e3:.2f} ms/step, {total} live rows, {B / dt:.0f} triplets/s")
