"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the flat
parameter layout is consistent, and the product refuses to run without a GPU."""
import ctypes as C

import pytest
import torch


def test_library_exports_every_header_symbol():
    from spn4cir_amd import _lib
    handle = _lib.lib()
    names = _lib.header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(handle, n), n
    assert set(names) == set(_lib._SIGS), set(names) ^ set(_lib._SIGS)
    assert handle.spn_abi_version() == 1
    assert handle.spn_error_string(-2).decode().startswith("unsupported shape")


def test_text_layout_matches_state_dict_spans():
    from spn4cir_amd import _lib, synthetic
    handle = _lib.lib()
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    cfg = _lib.TextCfg(256, 77, 77, W, heads, layers, D, 49408)
    lay = _lib.TextLayout()
    assert handle.spn_text_layout(C.byref(cfg), C.byref(lay)) == 0
    # 123.65 M trainable parameters of the ViT-L/14 text tower, minus logit_scale
    assert lay.n_params == 123650304
    assert lay.block_size == 12 * W * W + 13 * W
    assert lay.n_bf16 == layers * 24 * W * W + 2 * W * D
    assert handle.spn_text_act_bytes(C.byref(cfg)) > 6 * 2 ** 30 * 0.9
    # argument validation happens on the host, before any launch
    bad = _lib.TextCfg(4, 77, 77, 100, 2, 2, 64, 512)
    assert handle.spn_text_fwd(C.byref(bad), 1, 1, 1, 1, 1, None) == -2


def test_no_cpu_fallback():
    from spn4cir_amd import ops
    a = torch.zeros(128, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm_nt(a, a)
    if not torch.cuda.is_available():
        from spn4cir_amd.models import CIRPlus
        with pytest.raises(RuntimeError, match="no CPU path"):
            CIRPlus({"ln_final.weight": torch.zeros(64)}, device=torch.device("cpu"))


def test_synthetic_inputs_are_reproducible():
    from spn4cir_amd import synthetic
    ids = synthetic.token_ids(8, seed=1)
    assert ids.dtype == torch.int32 and ids.shape == (8, 77)
    assert (ids.argmax(dim=-1) == (ids != 0).sum(-1) - 1).all()       # EOT is the row maximum and the last token
    assert torch.equal(ids, synthetic.token_ids(8, seed=1))
    t, r = synthetic.banks(100, 64)
    assert torch.allclose(t.norm(dim=1), torch.ones(100), atol=1e-5)


def test_shard_range_partitions_exactly():
    from spn4cir_amd.distributed import shard_range
    for total, world in [(40000, 8), (10, 3), (7, 8), (100000, 8)]:
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
