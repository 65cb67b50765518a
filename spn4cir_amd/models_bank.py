"""zscir/models_bank.py `CIRPlus` protocol over the same MI355X path as spn4cir_amd.models.CIRPlus.

Differences of the zscir variant (zscir/models_bank.py:19-23,90-101,124-128): constructor argument order with
`combiner`, `label_smoothing`, `use_bank`; the reference row is always `refer_bank[indexs]` (per-triplet bank,
the `plus=False` behaviour); `forward(refer_image, text, target_image, indexs, target_indexs, refer_indexs,
grad_ckpt=False)` ignores both images when the banks are in use; `element_wise_sum(..., need_norm=False)`."""
import torch

from . import models as _m


class CIRPlus(_m.CIRPlus):
    def __init__(self, clip_model_name, combiner="sum", tau=0.01, label_smoothing=0, use_bank=False,
                 transform="targetpad", target_ratio=1.25, device=torch.device("cuda"), **kw):
        if combiner != "sum":
            raise ValueError("only the 'sum' combiner exists in the reference (models_bank.py:31-32)")
        super().__init__(clip_model_name, tau=tau, transform=transform, target_ratio=target_ratio, device=device,
                         plus=False, label_smoothing=label_smoothing, **kw)
        self.use_bank = use_bank

    def element_wise_sum(self, refer_image_feats, text_feats, need_norm=False):
        if need_norm:                                   # models_bank.py:49-54
            refer_image_feats = torch.nn.functional.normalize(refer_image_feats)
            text_feats = torch.nn.functional.normalize(text_feats)
        return refer_image_feats + text_feats

    element_wise_sum._spn_fused_sum = True            # with the default need_norm=False it IS the plain sum (validate._predict)

    def forward(self, refer_image, text, target_image, indexs, target_indexs, refer_indexs, grad_ckpt=False):
        return super().forward(text, indexs, target_indexs, refer_indexs)
