"""clip4cir/models_negtype.py on the MI355X path: the negative-type ablation of the in-batch step.

Same protocol as the reference class (models_negtype.py:19-38, :94-128): `CIRPlus(clip_model_name, tau, transform,
target_ratio, device, neg_type)`, `forward(text, indexs, target_indexs, refer_indexs, refer_image=, target_image=) ->
{'bbc_loss': 0-dim tensor with grad}`; both towers trainable.  The towers are the config-1 training towers
(spn4cir_amd.models.CIRPlus(wo_bank=True): spn_vision_fwd_train / spn_text_fwd and their backward passes; activations are
kept instead of recomputed under torch.utils.checkpoint), the four B x B InfoNCE terms and their feature gradients are ONE
C-ABI call (spn_negtype_head, csrc/negtype.hip) instead of the reference's Python loops over the batch (:53-80)."""
import torch

from . import gradsink, ops
from .models import CIRPlus as _CIRPlus


class CIRPlus(_CIRPlus):
    def __init__(self, clip_model_name, tau=0.01, transform="targetpad", target_ratio=1.25, device=torch.device("cuda"),
                 neg_type=4, **kw):
        """neg_type: bit mask 1..15 (models_negtype.py:108-127): 8 = query-negative, 4 = target-negative (the ordinary in-batch
        loss), 2 = text-negative, 1 = reference-negative term; the loss is the mean of the selected terms."""
        if not 1 <= int(neg_type) <= 15:
            raise ValueError("neg_type must be a bit mask in 1..15")
        super().__init__(clip_model_name, tau=tau, transform=transform, target_ratio=target_ratio, device=device, wo_bank=True,
                         **kw)
        self.neg_type = int(neg_type)

    def _inbatch_forward(self, ids, refer_image, target_image):
        if self.vision is None:
            raise RuntimeError("the negative-type step needs a ViT image tower")
        B = ids.shape[0]
        imgs = torch.cat([refer_image, target_image]).to(self.device, torch.float32)
        img_feats = self.vision.forward_train(imgs)                       # models_negtype.py:99-100 (both batches as one)
        text_feats = self.tower.forward(ids, *self._pack)                 # :98
        loss, d_refer, d_text, d_target = ops.negtype_head(img_feats[:B].contiguous(), text_feats.contiguous(),
                                                           img_feats[B:].contiguous(), self.tau, self.neg_type)
        return dict(loss=loss, d_refer=d_refer, d_text=d_text, d_target=d_target, B=B)

    def _inbatch_backward(self, st, grad_out):
        scale = self._dev_scale(grad_out, self.device)                    # d(loss) from autograd / GradScaler, on the device
        snap_t = gradsink.snapshot(self._params, self.tower.grads, self.tower.named_views)
        snap_v = gradsink.snapshot(self._params, self.vision.grads, self.vision.named_views, "visual.")
        flat_t = self.tower.backward(st["d_text"] * scale)
        flat_v = self.vision.backward(torch.cat([st["d_refer"], st["d_target"]]) * scale)
        gradsink.publish(self._params, flat_t, self.tower.named_views, snap_t)
        gradsink.publish(self._params, flat_v, self.vision.named_views, snap_v, "visual.")
