"""Run the BLIP trainer step with every scratch allocation poisoned (SPN_DEBUG_POISON=1) and report which gradient spans
come out non-finite (= some kernel read scratch it never wrote)."""
import os, sys, numpy as np, torch
os.environ["SPN_DEBUG_POISON"] = "1"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_fusion_gpu as T
gd = os.path.join(os.getcwd(), "tests", "golden")
from spn4cir_amd.fusion import BlipStage2Trainer
z, enc = T._blip_setup(gd)
tr = BlipStage2Trainer(enc, tau=float(z["tau"]), lr=1e-3)
tr.set_bank(torch.from_numpy(z["bank"]))
l = tr.step(torch.from_numpy(z["ids"]).cuda(), torch.from_numpy(z["mask"]).cuda(), torch.from_numpy(z["enc"]).cuda(), torch.from_numpy(z["labels"]).cuda())
print("loss", l.item(), "tau", tr.tau.item())
g = enc.grads.cpu()
bad = []
for k, off, shape in enc.spans():
    n = int(np.prod(shape))
    if not torch.isfinite(g[off:off + n]).all():
        bad.append((k, int((~torch.isfinite(g[off:off + n])).sum()), n))
print("non-finite gradient spans:", bad[:40], len(bad))
