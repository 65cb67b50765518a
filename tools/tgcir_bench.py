"""TG-CIR second-stage step timing on one GPU: CLIP ViT-B/16 text tower (W = 512, 12 layers) with
ln_final of all 77 positions, text_fc + TokenLearner + gated fusion head, bank InfoNCE over M = 40 000 x 512,
forward + backward (no optimizer; the reference's torch AdamW would run on the exposed parameters)."""
import sys, time, torch
sys.path.insert(0, ".")
from oracle import clip_text, tgcir_head
from spn4cir_amd.tgcir_models import CIRPlus

def main():
    B, M, C = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 40000, 512
    sd = clip_text.synthetic_text_state_dict(C, 12, C, seed=0)
    m = CIRPlus(sd, tau=0.02, plus=True)
    m.load_head(tgcir_head.synthetic_head(C, 8, 4, seed=1))
    g = torch.Generator().manual_seed(2)
    m.refer_bank = torch.randn(M, 12, C, generator=g).cuda() * 0.5
    m.target_bank = torch.nn.functional.normalize(torch.randn(M, C, generator=g), dim=-1)
    ids = clip_text.synthetic_token_ids(B, seed=1).cuda()
    ridx = torch.randint(0, M, (B,), generator=g).cuda()
    labels = torch.randint(0, M, (B,), generator=g)
    def step():
        for p in m.parameters():
            p.grad = None
        loss = m.forward(ids, None, labels, ridx)["bank_loss"]
        loss.backward()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 10
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"TG-CIR step B={B}: {dt*1e3:.2f} ms  {B/dt:.0f} triplets/s  loss {loss.item():.3f}")

main()
