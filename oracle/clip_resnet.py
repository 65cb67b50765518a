"""CPU oracle of CLIP's ModifiedResNet image tower (test infrastructure only).

clip4cir/clip/model.py:10-155: three-convolution stem + AvgPool2d(2); Bottleneck blocks whose strided convolutions are
replaced by AvgPool2d(stride) after conv2 (and in front of the 1x1 downsample convolution); AttentionPool2d (mean token
prepended, positional embedding, one multi-head attention with the mean token as the only query).  Eval-mode
BatchNorm (running statistics), as the frozen tower runs in stage 2."""
import torch
import torch.nn.functional as F


def resnet_cfg_from_state_dict(sd, prefix="visual."):
    """clip/model.py:412-419."""
    counts = [len({k.split(".")[2] for k in sd if k.startswith(f"{prefix}layer{b}.")}) for b in (1, 2, 3, 4)]
    width = sd[prefix + "layer1.0.conv1.weight"].shape[0]
    grid = round((sd[prefix + "attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
    return dict(layers=tuple(counts), width=width, res=grid * 32, heads=width * 32 // 64,
                embed_dim=sd[prefix + "attnpool.c_proj.weight"].shape[0])


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _bottleneck(x, sd, p, stride):
    out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    out = F.relu(_bn(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2."))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3.")
    identity = x
    if p + "downsample.0.weight" in sd:
        identity = F.avg_pool2d(x, stride) if stride > 1 else x
        identity = _bn(F.conv2d(identity, sd[p + "downsample.0.weight"]), sd, p + "downsample.1.")
    return F.relu(out + identity)


def encode_image(sd, image, prefix="visual."):
    sd = {k[len(prefix):]: v.float() for k, v in sd.items() if k.startswith(prefix)}
    cfg_layers = [len({k.split(".")[1] for k in sd if k.startswith(f"layer{b}.")}) for b in (1, 2, 3, 4)]
    x = image.float()
    x = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=1), sd, "bn1."))
    x = F.relu(_bn(F.conv2d(x, sd["conv2.weight"], padding=1), sd, "bn2."))
    x = F.relu(_bn(F.conv2d(x, sd["conv3.weight"], padding=1), sd, "bn3."))
    x = F.avg_pool2d(x, 2)
    for li, n in enumerate(cfg_layers, start=1):
        for bi in range(n):
            x = _bottleneck(x, sd, f"layer{li}.{bi}.", 2 if (bi == 0 and li > 1) else 1)
    B, C, H, W = x.shape
    t = x.flatten(2).permute(2, 0, 1)                                   # (HW) B C
    t = torch.cat([t.mean(dim=0, keepdim=True), t], dim=0) + sd["attnpool.positional_embedding"][:, None, :]
    heads = C // 64
    q = F.linear(t[:1], sd["attnpool.q_proj.weight"], sd["attnpool.q_proj.bias"])
    k = F.linear(t, sd["attnpool.k_proj.weight"], sd["attnpool.k_proj.bias"])
    v = F.linear(t, sd["attnpool.v_proj.weight"], sd["attnpool.v_proj.bias"])
    S = t.shape[0]
    qh = q.reshape(1, B, heads, 64).permute(1, 2, 0, 3) * (64 ** -0.5)
    kh = k.reshape(S, B, heads, 64).permute(1, 2, 0, 3)
    vh = v.reshape(S, B, heads, 64).permute(1, 2, 0, 3)
    a = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1) @ vh             # B heads 1 64
    o = a.permute(2, 0, 1, 3).reshape(B, C)
    return F.linear(o, sd["attnpool.c_proj.weight"], sd["attnpool.c_proj.bias"])
