"""The reference's own loop body (train_negplus.py:107-123) on the drop-in CIRPlus: captions as strings (host BPE every
step), autograd backward, torch.optim.AdamW on the exposed parameters, GradScaler as in the reference - vs the fused
Stage2Trainer path that bench.py times.  Config 2 shape."""
import os, random, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd import synthetic
from spn4cir_amd.models import CIRPlus

def run(fused_optim=False, steps=10, warmup=3):
    B, M = 256, 40000
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    model = CIRPlus(sd, tau=0.02, device=torch.device("cuda"), plus=True)
    target, refer = synthetic.banks(M, D, seed=2)
    model.refer_bank, model.target_bank = refer, target
    random.seed(0)
    words = ("is red and has long sleeves with a floral pattern shorter more colorful dress shirt top blue striped darker "
             "lighter green collar buttons casual formal the same but different style material lace silk cotton").split()
    caps = [" ".join(random.choice(words) for _ in range(random.randint(5, 30))) for _ in range(B)]
    ridx, labels = synthetic.triplet_indices(B, M, seed=4)
    if fused_optim:                          # the one-line change of INTEGRATION.md: same arguments, one fused launch
        from spn4cir_amd.optim import AdamW
        opt = AdamW(model.parameters(), lr=2e-5, betas=(0.9, 0.999), eps=1e-7)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=2e-5, betas=(0.9, 0.999), eps=1e-7)
    scaler = torch.cuda.amp.GradScaler()
    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.forward(caps, None, labels, ridx)["bank_loss"]
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update()
        model.parameters_changed()
        return loss
    for _ in range(warmup): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = steps
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    return {"optimizer": "spn4cir_amd.optim.AdamW" if fused_optim else "torch.optim.AdamW", "ms_per_step": round(dt * 1e3, 3),
            "triplets_per_s": round(B / dt, 1), "loss_last": round(float(loss.item()), 5)}


if __name__ == "__main__":
    r = run("--fused-optim" in sys.argv)
    print(f"drop-in loop ({r['optimizer']}): {r['ms_per_step']:.2f} ms/step  {r['triplets_per_s']:.0f} triplets/s  loss {r['loss_last']:.4f}")
