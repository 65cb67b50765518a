"""Cycle stamps of the NT epilogue's sub-phases (mid-grid workgroup, wave 0).  Needs the probe build:
    tools/build_variant.sh probes "-DSPN_GEMM_PROBES"
    SPN_LIB_PATH=spn4cir_amd/libspn4cir_hip_probes.so SPN_GEMM_DBG=192 python tools/epi_probe.py
Output bytes are overwritten by design.  Stamps: barrier 1, LDS writes, barrier 2, read-back + stores (chunk 0), then the same for chunk 1, store drain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spn4cir_amd import ops
for (M, N, K) in [(19712, 3072, 768), (19712, 768, 768)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda")
    for _ in range(3): ops.gemm_nt(a, b, bias)
    for trial in range(3):
        out = ops.gemm_nt(a, b, bias); torch.cuda.synchronize()
        c = out.view(-1)[:40].view(torch.int32).cpu().tolist()
        print(M, N, K, "phases(setup,prologue,kloop,epi)", [x & 0xffffffff for x in c[:4]], "epi stamps", [x & 0xffffffff for x in c[4:16]])
