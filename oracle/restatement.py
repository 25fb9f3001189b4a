"""TEST INFRASTRUCTURE ONLY -- the CPU oracle for the MI355X hot path. Never imported by generativemodels_amd/.

A *functional* restatement (plain torch CPU ops over a reference-keyed ``state_dict`` + the constructor kwargs) of the
reference's algorithm for the hot path of SURVEY.md section 8(a): DiffusionModelUNet forward, AutoencoderKL / VQVAE
encode-decode, the DDPM / DDIM scheduler arithmetic and the DiffusionInferer / LatentDiffusionInferer sampling loops.
Every function cites the reference file:line it follows (paths relative to /root/reference/generative/).

Pinning: the reference's own tests hold no golden tensors for this path (shape-only), so this
restatement is pinned against outputs of the UNMODIFIED reference executed in the build container through
oracle/monai_stub.py: tests/golden/*.pt (made by oracle/make_golden.py, committed) and, where /root/reference exists,
live comparisons in tests/test_oracle_vs_reference.py. It travels to the GPU box (the reference does not) and is the
checker for the `-m gpu` parity tests, smoke() and bench.py's `cpu_baseline` leg (kind "port").

It works in whatever float dtype the inputs / state_dict are given in (fp32 for the timed CPU baseline, fp64 for
tolerance work).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------------------------------


def _rep(v, n):
    """monai.utils.ensure_tuple_rep as used at networks/nets/diffusion_model_unet.py:1714,1723."""
    if isinstance(v, (list, tuple)):
        if len(v) != n:
            raise ValueError(f"Sequence must have length {n}, got {len(v)}.")
        return tuple(v)
    return (v,) * n


def _convnd(x, w, b, stride=1, padding=0, dilation=1):
    fn = {3: F.conv1d, 4: F.conv2d, 5: F.conv3d}[x.ndim]
    return fn(x, w, b, stride=stride, padding=padding, dilation=dilation)


def _convtnd(x, w, b, stride, padding, output_padding, dilation=1):
    fn = {3: F.conv_transpose1d, 4: F.conv_transpose2d, 5: F.conv_transpose3d}[x.ndim]
    return fn(x, w, b, stride=stride, padding=padding, output_padding=output_padding, dilation=dilation)


def _conv(sd, p, x, stride=1, padding=1, dilation=1):
    """MONAI Convolution(conv_only=True): child `conv` = nn.ConvNd (nets/diffusion_model_unet.py:1748-1756 et al.)."""
    return _convnd(x, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), stride, padding, dilation)


def _gn(sd, p, x, groups, eps):
    """nn.GroupNorm(groups, C, eps, affine=True): nets/diffusion_model_unet.py:623,643,275,377,1854."""
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _up2(x):
    """F.interpolate(scale_factor=2, mode="nearest"): nets/diffusion_model_unet.py:578 / nets/autoencoderkl.py:88."""
    return F.interpolate(x, scale_factor=2.0, mode="nearest")


def _avgpool2(x):
    fn = {4: F.avg_pool2d, 5: F.avg_pool3d}[x.ndim]
    return fn(x, kernel_size=2, stride=2)


def _tokens(x):
    """(N, C, *sp) -> (N, L, C), row-major over the spatial dims (nets/diffusion_model_unet.py:430-433,328-331)."""
    n, c = x.shape[:2]
    return x.reshape(n, c, -1).transpose(1, 2)


def _untokens(t, like):
    """(N, L, C) -> (N, C, *sp) (nets/diffusion_model_unet.py:453-456,336-339)."""
    n, c = t.shape[0], t.shape[2]
    return t.transpose(1, 2).reshape(n, c, *like.shape[2:])


def _mha(q, k, v, heads, scale, upcast=False):
    """softmax(scale * Q K^T) V per head (nets/diffusion_model_unet.py:117-153 and 387-415).

    q: (N, Lq, H*d); k, v: (N, Lk, H*d). `upcast` follows CrossAttention._attention:139-151 (q, k to fp32, probs back).
    """
    n, lq, inner = q.shape
    d = inner // heads

    def split(t):
        return t.reshape(n, t.shape[1], heads, d).permute(0, 2, 1, 3)

    qh, kh, vh = split(q), split(k), split(v)
    dt = qh.dtype
    if upcast:
        qh, kh = qh.float(), kh.float()
    probs = (scale * (qh @ kh.transpose(-1, -2))).softmax(dim=-1).to(dt)
    return (probs @ vh).permute(0, 2, 1, 3).reshape(n, lq, inner)


# --------------------------------------------------------------------------------------------------------------------
# DiffusionModelUNet (networks/nets/diffusion_model_unet.py)
# --------------------------------------------------------------------------------------------------------------------

UNET_DEFAULTS = dict(
    num_res_blocks=(2, 2, 2, 2), num_channels=(32, 64, 64, 64), attention_levels=(False, False, True, True),
    norm_num_groups=32, norm_eps=1e-6, resblock_updown=False, num_head_channels=8, with_conditioning=False,
    transformer_num_layers=1, cross_attention_dim=None, num_class_embeds=None, upcast_attention=False,
    use_flash_attention=False, dropout_cattn=0.0,
)  # nets/diffusion_model_unet.py:1673-1692


def timestep_embedding(timesteps, dim, max_period=10000):
    """get_timestep_embedding, nets/diffusion_model_unet.py:461-485: [cos | sin], zero-pad if dim is odd."""
    if timesteps.ndim != 1:
        raise ValueError("Timesteps should be a 1d-array")
    half = dim // 2
    expo = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    freqs = torch.exp(expo / half)
    args = timesteps[:, None].float() * freqs[None, :]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def unet_resnet(sd, p, x, emb, groups, eps, up=False, down=False):
    """ResnetBlock.forward, nets/diffusion_model_unet.py:669-696 (up/down resamples BOTH x and h: 674-682)."""
    h = F.silu(_gn(sd, p + ".norm1", x, groups, eps))
    if up:
        x, h = _up2(x), _up2(h)
    elif down:
        x, h = _avgpool2(x), _avgpool2(h)
    h = _conv(sd, p + ".conv1", h)
    t = _lin(sd, p + ".time_emb_proj", F.silu(emb))
    h = h + t.reshape(*t.shape, *([1] * (x.ndim - 2)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups, eps)))
    if (p + ".skip_connection.conv.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def spade(sd, p, x, seg, groups, eps):
    """SPADE.forward, networks/blocks/spade_norm.py:79-96.  The parameter-free norm is GroupNorm (affine iff `param_free_norm.N.weight`
    is in the state dict: spade_diffusion_model_unet.py:110-118 vs spade_autoencoderkl.py:72-80); mlp_shared is conv + LeakyReLU(0.01);
    mlp_gamma / mlp_beta are MONAI Convolution(act=None) and therefore conv + the default InstanceNorm (affine-free, eps 1e-5)."""
    normalized = F.group_norm(x, groups, sd.get(p + ".param_free_norm.N.weight"), sd.get(p + ".param_free_norm.N.bias"), eps)
    segmap = F.interpolate(seg, size=x.shape[2:], mode="nearest")
    k = sd[p + ".mlp_shared.conv.weight"].shape[-1]
    actv = F.leaky_relu(_conv(sd, p + ".mlp_shared", segmap, padding=k // 2), 0.01)
    gamma = F.instance_norm(_conv(sd, p + ".mlp_gamma", actv, padding=k // 2), eps=1e-5)
    beta = F.instance_norm(_conv(sd, p + ".mlp_beta", actv, padding=k // 2), eps=1e-5)
    return normalized * (1 + gamma) + beta


def spade_unet_resnet(sd, p, x, emb, seg, groups, eps):
    """SPADEResnetBlock.forward, nets/spade_diffusion_model_unet.py:173-200 (the decoder blocks are never up / down sampling)."""
    h = _conv(sd, p + ".conv1", F.silu(spade(sd, p + ".norm1", x, seg, groups, eps)))
    t = _lin(sd, p + ".time_emb_proj", F.silu(emb))
    h = h + t.reshape(*t.shape, *([1] * (x.ndim - 2)))
    h = _conv(sd, p + ".conv2", F.silu(spade(sd, p + ".norm2", h, seg, groups, eps)))
    if (p + ".skip_connection.conv.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def spade_aekl_resblock(sd, p, x, seg, groups):
    """SPADEResBlock.forward, nets/spade_autoencoderkl.py:122-134; its GroupNorm is built without `eps`: nn.GroupNorm's default 1e-5."""
    h = _conv(sd, p + ".conv1", F.silu(spade(sd, p + ".norm1", x, seg, groups, 1e-5)))
    h = _conv(sd, p + ".conv2", F.silu(spade(sd, p + ".norm2", h, seg, groups, 1e-5)))
    if (p + ".nin_shortcut.conv.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def attention_block(sd, p, x, groups, eps, num_head_channels):
    """AttentionBlock.forward, nets/diffusion_model_unet.py:418-458 (twin nets/autoencoderkl.py:272-312).

    proj_attn exists in the state_dict but is never applied (constructed at :383, absent from forward)."""
    c = x.shape[1]
    heads = c // num_head_channels if num_head_channels is not None else 1
    scale = 1 / math.sqrt(c / heads)
    t = _tokens(_gn(sd, p + ".norm", x, groups, eps))
    o = _mha(_lin(sd, p + ".to_q", t), _lin(sd, p + ".to_k", t), _lin(sd, p + ".to_v", t), heads, scale)
    return _untokens(o, x) + x


def cross_attention(sd, p, x, context, heads, head_ch, upcast):
    """CrossAttention.forward, nets/diffusion_model_unet.py:155-175 (q/k/v bias-free: 106-108; to_out.0 Linear: 110)."""
    ctx = x if context is None else context
    o = _mha(_lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", ctx), _lin(sd, p + ".to_v", ctx), heads,
             1 / math.sqrt(head_ch), upcast)
    return _lin(sd, p + ".to_out.0", o)


def transformer_block(sd, p, x, context, heads, head_ch, upcast):
    """BasicTransformerBlock.forward, nets/diffusion_model_unet.py:225-234; ff = MLPBlock(act="GEGLU") (:211)."""
    c = x.shape[-1]

    def ln(name, t):
        return F.layer_norm(t, (c,), sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], 1e-5)

    x = cross_attention(sd, p + ".attn1", ln("norm1", x), None, heads, head_ch, upcast) + x
    x = cross_attention(sd, p + ".attn2", ln("norm2", x), context, heads, head_ch, upcast) + x
    a, gate = _lin(sd, p + ".ff.linear1", ln("norm3", x)).chunk(2, dim=-1)  # GEGLU: x * gelu(gate), exact erf GELU
    return _lin(sd, p + ".ff.linear2", a * F.gelu(gate)) + x


def spatial_transformer(sd, p, x, context, groups, eps, num_head_channels, num_layers, upcast):
    """SpatialTransformer.forward, nets/diffusion_model_unet.py:314-342."""
    heads = x.shape[1] // num_head_channels
    h = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, groups, eps), padding=0)
    t = _tokens(h)
    for i in range(num_layers):
        t = transformer_block(sd, f"{p}.transformer_blocks.{i}", t, context, heads, num_head_channels, upcast)
    h = _conv(sd, p + ".proj_out", _untokens(t, h), padding=0)
    return h + x


def unet_forward(sd, cfg, x, timesteps, context=None, class_labels=None, down_block_additional_residuals=None,
                 mid_block_additional_residual=None, seg=None):
    """DiffusionModelUNet.forward, nets/diffusion_model_unet.py:1869-1943, over the block layout built at :1770-1867.

    cfg: the constructor kwargs (spatial_dims, in_channels, out_channels + any of UNET_DEFAULTS)."""
    c = dict(UNET_DEFAULTS, **{k: v for k, v in cfg.items() if k not in ("label_nc", "spade_intermediate_channels")})
    # seg given: SPADEDiffusionModelUNet.forward (nets/spade_diffusion_model_unet.py:836-912) -- same network, SPADE decoder resnets
    chans = tuple(c["num_channels"])
    nlev = len(chans)
    att = tuple(c["attention_levels"])
    nres = _rep(c["num_res_blocks"], nlev)
    nhc = _rep(c["num_head_channels"], nlev)
    groups, eps = c["norm_num_groups"], c["norm_eps"]
    cond, updown = c["with_conditioning"], c["resblock_updown"]
    nlayers, upcast = c["transformer_num_layers"], c["upcast_attention"]

    # 1. time (+ class) embedding: :1888-1902
    t_emb = timestep_embedding(timesteps, chans[0]).to(dtype=x.dtype)
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", t_emb)))
    if c["num_class_embeds"] is not None:
        if class_labels is None:
            raise ValueError("class_labels should be provided when num_class_embeds > 0")
        emb = emb + F.embedding(class_labels, sd["class_embedding.weight"]).to(dtype=x.dtype)
    if context is not None and not cond:
        raise ValueError("model should have with_conditioning = True if context is provided")

    def attend(p, h, level_nhc):
        if cond:
            return spatial_transformer(sd, p, h, context, groups, eps, level_nhc, nlayers, upcast)
        return attention_block(sd, p, h, groups, eps, level_nhc)

    # 3./4. conv_in and down path: :1905-1914 -> Down/AttnDown/CrossAttnDown blocks :699-1010
    h = _conv(sd, "conv_in", x)
    skips = [h]
    for i in range(nlev):
        p = f"down_blocks.{i}"
        for j in range(nres[i]):
            h = unet_resnet(sd, f"{p}.resnets.{j}", h, emb, groups, eps)
            if att[i]:
                h = attend(f"{p}.attentions.{j}", h, nhc[i])
            skips.append(h)
        if i != nlev - 1:
            if updown:
                h = unet_resnet(sd, f"{p}.downsampler", h, emb, groups, eps, down=True)
            else:
                h = _conv(sd, f"{p}.downsampler.op", h, stride=2, padding=1)  # Downsample :510-518
            skips.append(h)
    if down_block_additional_residuals is not None:  # :1917-1925
        skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]

    # 5. mid: :1928 -> AttnMidBlock/CrossAttnMidBlock :1013-1148 (always has attention)
    h = unet_resnet(sd, "middle_block.resnet_1", h, emb, groups, eps)
    h = attend("middle_block.attention", h, nhc[-1])
    h = unet_resnet(sd, "middle_block.resnet_2", h, emb, groups, eps)
    if mid_block_additional_residual is not None:
        h = h + mid_block_additional_residual

    # 6. up path: :1935-1938 -> Up/AttnUp/CrossAttnUp blocks :1151-1469 (num_res_blocks+1 resnets, LIFO skips)
    ratt, rres, rnhc = att[::-1], nres[::-1], nhc[::-1]
    for i in range(nlev):
        p = f"up_blocks.{i}"
        for j in range(rres[i] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            if seg is None:
                h = unet_resnet(sd, f"{p}.resnets.{j}", h, emb, groups, eps)
            else:
                h = spade_unet_resnet(sd, f"{p}.resnets.{j}", h, emb, seg, groups, eps)
            if ratt[i]:
                h = attend(f"{p}.attentions.{j}", h, rnhc[i])
        if i != nlev - 1:
            if updown:
                h = unet_resnet(sd, f"{p}.upsampler", h, emb, groups, eps, up=True)
            else:
                h = _conv(sd, f"{p}.upsampler.conv", _up2(h))  # Upsample :572-585
    # 7. out head GN -> SiLU -> conv: :1853-1867, :1941
    return _conv(sd, "out.2", F.silu(_gn(sd, "out.0", h, groups, eps)))


# --------------------------------------------------------------------------------------------------------------------
# ControlNet (networks/nets/controlnet.py)
# --------------------------------------------------------------------------------------------------------------------

CONTROLNET_DEFAULTS = dict(UNET_DEFAULTS, conditioning_embedding_in_channels=1, conditioning_embedding_num_channels=(16, 32, 96, 256))


def controlnet_cond_embedding(sd, p, cond, num_channels):
    """ControlNetConditioningEmbedding.forward, nets/controlnet.py:103-114: conv_in, (conv, stride-2 conv) per level, SiLU after each,
    zero-initialised conv_out."""
    e = F.silu(_conv(sd, f"{p}.conv_in", cond))
    for i in range(len(num_channels) - 1):
        e = F.silu(_conv(sd, f"{p}.blocks.{2 * i}", e))
        e = F.silu(_conv(sd, f"{p}.blocks.{2 * i + 1}", e, stride=2, padding=1))
    return _conv(sd, f"{p}.conv_out", e)


def controlnet_forward(sd, cfg, x, timesteps, controlnet_cond, conditioning_scale=1.0, context=None, class_labels=None):
    """ControlNet.forward, nets/controlnet.py:367-436: the UNet's encoder + mid block on x + embed(cond), every skip and the mid
    output through its own zero-initialised 1x1 conv, all scaled.  -> (tuple of down residuals, mid residual)."""
    c = dict(CONTROLNET_DEFAULTS, **cfg)
    chans = tuple(c["num_channels"])
    nlev = len(chans)
    att = tuple(c["attention_levels"])
    nres = _rep(c["num_res_blocks"], nlev)
    nhc = _rep(c["num_head_channels"], nlev)
    groups, eps = c["norm_num_groups"], c["norm_eps"]
    cond, updown = c["with_conditioning"], c["resblock_updown"]
    nlayers, upcast = c["transformer_num_layers"], c["upcast_attention"]
    t_emb = timestep_embedding(timesteps, chans[0]).to(dtype=x.dtype)
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", t_emb)))
    if c["num_class_embeds"] is not None:
        if class_labels is None:
            raise ValueError("class_labels should be provided when num_class_embeds > 0")
        emb = emb + F.embedding(class_labels, sd["class_embedding.weight"]).to(dtype=x.dtype)
    if context is not None and not cond:
        raise ValueError("model should have with_conditioning = True if context is provided")

    def attend(p, h, level_nhc):
        if cond:
            return spatial_transformer(sd, p, h, context, groups, eps, level_nhc, nlayers, upcast)
        return attention_block(sd, p, h, groups, eps, level_nhc)

    h = _conv(sd, "conv_in", x) + controlnet_cond_embedding(sd, "controlnet_cond_embedding", controlnet_cond,
                                                              c["conditioning_embedding_num_channels"])
    skips = [h]
    for i in range(nlev):
        p = f"down_blocks.{i}"
        for j in range(nres[i]):
            h = unet_resnet(sd, f"{p}.resnets.{j}", h, emb, groups, eps)
            if att[i]:
                h = attend(f"{p}.attentions.{j}", h, nhc[i])
            skips.append(h)
        if i != nlev - 1:
            if updown:
                h = unet_resnet(sd, f"{p}.downsampler", h, emb, groups, eps, down=True)
            else:
                h = _conv(sd, f"{p}.downsampler.op", h, stride=2, padding=1)
            skips.append(h)
    h = unet_resnet(sd, "middle_block.resnet_1", h, emb, groups, eps)
    h = attend("middle_block.attention", h, nhc[-1])
    h = unet_resnet(sd, "middle_block.resnet_2", h, emb, groups, eps)
    outs = []
    for k, s_ in enumerate(skips):
        # the first zero conv is registered as the bare nn.Conv (controlnet.py:277-287), the others as Convolution wrappers
        name = f"controlnet_down_blocks.{k}" if k == 0 else f"controlnet_down_blocks.{k}.conv"
        outs.append(_convnd(s_, sd[name + ".weight"], sd[name + ".bias"]) * conditioning_scale)
    mid = _conv(sd, "controlnet_mid_block", h, padding=0) * conditioning_scale
    return tuple(outs), mid


# --------------------------------------------------------------------------------------------------------------------
# AutoencoderKL (networks/nets/autoencoderkl.py)
# --------------------------------------------------------------------------------------------------------------------

AEKL_DEFAULTS = dict(
    in_channels=1, out_channels=1, num_res_blocks=(2, 2, 2, 2), num_channels=(32, 64, 64, 64),
    attention_levels=(False, False, True, True), latent_channels=3, norm_num_groups=32, norm_eps=1e-6,
    with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True, use_flash_attention=False,
    use_checkpointing=False, use_convtranspose=False,
)  # nets/autoencoderkl.py:623-639


def aekl_resblock(sd, p, x, groups, eps):
    """ResBlock.forward, nets/autoencoderkl.py:180-193."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups, eps)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups, eps)))
    if (p + ".nin_shortcut.conv.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def _aekl_cfg(cfg):
    c = dict(AEKL_DEFAULTS, **cfg)
    c["num_res_blocks"] = _rep(c["num_res_blocks"], len(c["num_channels"]))
    return c


def aekl_encoder(sd, cfg, x):
    """Encoder, nets/autoencoderkl.py:355-453 (block list order) ; Downsample pads the HIGH side only (:107,120)."""
    c = _aekl_cfg(cfg)
    chans, att, nres = c["num_channels"], c["attention_levels"], c["num_res_blocks"]
    groups, eps = c["norm_num_groups"], c["norm_eps"]
    k = 0
    p = "encoder.blocks."
    h = _conv(sd, f"{p}{k}", x)
    k += 1
    for i in range(len(chans)):
        for _ in range(nres[i]):
            h = aekl_resblock(sd, f"{p}{k}", h, groups, eps)
            k += 1
            if att[i]:
                h = attention_block(sd, f"{p}{k}", h, groups, eps, None)
                k += 1
        if i != len(chans) - 1:
            h = F.pad(h, (0, 1) * (h.ndim - 2), mode="constant", value=0.0)
            h = _conv(sd, f"{p}{k}.conv", h, stride=2, padding=0)  # Downsample.conv = Convolution -> .conv.conv.*
            k += 1
    if c["with_encoder_nonlocal_attn"]:
        h = aekl_resblock(sd, f"{p}{k}", h, groups, eps)
        h = attention_block(sd, f"{p}{k + 1}", h, groups, eps, None)
        h = aekl_resblock(sd, f"{p}{k + 2}", h, groups, eps)
        k += 3
    h = _gn(sd, f"{p}{k}", h, groups, eps)  # NOTE: no SiLU between this norm and the last conv (:433-446)
    return _conv(sd, f"{p}{k + 1}", h)


def aekl_encode(sd, cfg, x):
    """AutoencoderKL.encode, nets/autoencoderkl.py:718-736 -> (z_mu, z_sigma)."""
    h = aekl_encoder(sd, cfg, x)
    z_mu = _conv(sd, "quant_conv_mu", h, padding=0)
    z_log_var = torch.clamp(_conv(sd, "quant_conv_log_sigma", h, padding=0), -30.0, 20.0)
    return z_mu, torch.exp(z_log_var / 2)


def aekl_decode(sd, cfg, z, seg=None):
    """AutoencoderKL.decode, nets/autoencoderkl.py:769-784 -> Decoder :500-597.  seg given: SPADEAutoencoderKL.decode,
    nets/spade_autoencoderkl.py:457-469 -> SPADEDecoder :137-289 (same layout, SPADE residual blocks)."""
    c = _aekl_cfg({k: v for k, v in cfg.items() if k not in ("label_nc", "spade_intermediate_channels")})
    res = aekl_resblock if seg is None else (lambda sd_, p_, h_, g_, e_: spade_aekl_resblock(sd_, p_, h_, seg, g_))
    chans, att, nres = c["num_channels"][::-1], c["attention_levels"][::-1], c["num_res_blocks"][::-1]
    groups, eps = c["norm_num_groups"], c["norm_eps"]
    h = _conv(sd, "post_quant_conv", z, padding=0)
    p = "decoder.blocks."
    k = 0
    h = _conv(sd, f"{p}{k}", h)
    k += 1
    if c["with_decoder_nonlocal_attn"]:
        h = res(sd, f"{p}{k}", h, groups, eps)
        h = attention_block(sd, f"{p}{k + 1}", h, groups, eps, None)
        h = res(sd, f"{p}{k + 2}", h, groups, eps)
        k += 3
    for i in range(len(chans)):
        for _ in range(nres[i]):
            h = res(sd, f"{p}{k}", h, groups, eps)
            k += 1
            if att[i]:
                h = attention_block(sd, f"{p}{k}", h, groups, eps, None)
                k += 1
        if i != len(chans) - 1:
            if c["use_convtranspose"]:  # Upsample :54-63: ConvTranspose k3 s2 p1 output_padding=1
                h = _convtnd(h, sd[f"{p}{k}.conv.conv.weight"], sd[f"{p}{k}.conv.conv.bias"], 2, 1, 1)
            else:
                h = _convnd(_up2(h), sd[f"{p}{k}.conv.conv.weight"], sd[f"{p}{k}.conv.conv.bias"], 1, 1)
            k += 1
    h = _gn(sd, f"{p}{k}", h, groups, eps)
    return _conv(sd, f"{p}{k + 1}", h)


# --------------------------------------------------------------------------------------------------------------------
# VQVAE + quantizer (networks/nets/vqvae.py, networks/layers/vector_quantizer.py)
# --------------------------------------------------------------------------------------------------------------------

VQVAE_DEFAULTS = dict(
    num_channels=(96, 96, 192), num_res_layers=3, num_res_channels=(96, 96, 192),
    downsample_parameters=((2, 4, 1, 1), (2, 4, 1, 1), (2, 4, 1, 1)),
    upsample_parameters=((2, 4, 1, 1, 0), (2, 4, 1, 1, 0), (2, 4, 1, 1, 0)), num_embeddings=32, embedding_dim=64,
    embedding_init="normal", commitment_cost=0.25, decay=0.5, epsilon=1e-5, dropout=0.0, act="RELU", output_act=None,
    ddp_sync=True, use_checkpointing=False,
)  # nets/vqvae.py:303-326

_ACTS = {"relu": F.relu, "leakyrelu": F.leaky_relu, "gelu": F.gelu, "silu": F.silu, "swish": F.silu,
         "tanh": torch.tanh, "sigmoid": torch.sigmoid, "elu": F.elu}


def _vq_cfg(cfg):
    c = dict(VQVAE_DEFAULTS, **cfg)
    n = len(c["num_channels"])
    c["num_res_channels"] = _rep(c["num_res_channels"], n)
    if all(isinstance(v, int) for v in c["downsample_parameters"]):
        c["downsample_parameters"] = (tuple(c["downsample_parameters"]),) * n
    if all(isinstance(v, int) for v in c["upsample_parameters"]):
        c["upsample_parameters"] = (tuple(c["upsample_parameters"]),) * n
    return c


def vqvae_res_unit(sd, p, x, act):
    """VQVAEResidualUnit.forward, nets/vqvae.py:79-80: relu(x + conv2(act(conv1(x)))) (ADN "DA": no norm applied)."""
    h = act(_conv(sd, p + ".conv1", x))
    return F.relu(x + _conv(sd, p + ".conv2", h))


def vqvae_encode(sd, cfg, x):
    """VQVAE.encode -> Encoder, nets/vqvae.py:124-171,414-421 (eval mode: dropout is the identity)."""
    c = _vq_cfg(cfg)
    act = _ACTS[str(c["act"]).lower()]
    k = 0
    h = x
    for i in range(len(c["num_channels"])):
        s, ks, d, pad = c["downsample_parameters"][i]
        h = act(_conv(sd, f"encoder.blocks.{k}", h, stride=s, padding=pad, dilation=d))
        k += 1
        for _ in range(c["num_res_layers"]):
            h = vqvae_res_unit(sd, f"encoder.blocks.{k}", h, act)
            k += 1
    return _conv(sd, f"encoder.blocks.{k}", h)


def vqvae_decode(sd, cfg, z):
    """VQVAE.decode -> Decoder, nets/vqvae.py:206-272."""
    c = _vq_cfg(cfg)
    act = _ACTS[str(c["act"]).lower()]
    n = len(c["num_channels"])
    k = 0
    h = _conv(sd, f"decoder.blocks.{k}", z)
    k += 1
    for i in range(n):
        for _ in range(c["num_res_layers"]):
            h = vqvae_res_unit(sd, f"decoder.blocks.{k}", h, act)
            k += 1
        s, ks, d, pad, opad = c["upsample_parameters"][i]
        h = _convtnd(h, sd[f"decoder.blocks.{k}.conv.weight"], sd.get(f"decoder.blocks.{k}.conv.bias"), s, pad, opad, d)
        if i != n - 1:
            h = act(h)
        k += 1
    if c["output_act"]:
        h = _ACTS[str(c["output_act"]).lower()](h)
    return h


def vq_index_quantize(sd, z):
    """EMAQuantizer.quantize, layers/vector_quantizer.py:86-122: fp32, ||x||^2 + ||e||^2 - 2 x.E^T, argmax(-d)."""
    emb = sd["quantizer.quantizer.embedding.weight"].float()
    z = z.float()
    perm = [0] + list(range(2, z.ndim)) + [1]
    flat = z.permute(perm).contiguous().view(-1, emb.shape[1])
    dist = (flat**2).sum(dim=1, keepdim=True) + (emb.t() ** 2).sum(dim=0, keepdim=True) - 2 * torch.mm(flat, emb.t())
    idx = torch.max(-dist, dim=1)[1]
    shape = list(z.shape)
    del shape[1]
    return idx.view(shape), dist


def vq_embed(sd, idx):
    """EMAQuantizer.embed, layers/vector_quantizer.py:124-138."""
    emb = sd["quantizer.quantizer.embedding.weight"]
    perm = [0, idx.ndim] + list(range(1, idx.ndim))
    return F.embedding(idx, emb).permute(perm).contiguous()


def vq_quantize(sd, cfg, z):
    """VQVAE.quantize (eval): layers/vector_quantizer.py:161-188,208-220 -> (quantized, loss)."""
    c = _vq_cfg(cfg)
    idx, _ = vq_index_quantize(sd, z)
    q = vq_embed(sd, idx)
    loss = c["commitment_cost"] * F.mse_loss(q, z)
    return q.to(z.dtype), loss


def vq_ema_forward(state, args, x):
    """EMAQuantizer.forward in train() mode, layers/vector_quantizer.py:161-188, functionally: -> (quantized, loss, indices, new state).
    state: dict(embedding.weight, ema_cluster_size, ema_w) (not modified); args: num_embeddings, embedding_dim, commitment_cost, decay,
    epsilon.  The lookup (:162-163) precedes the EMA update (:166-180); quantized / loss are differentiable in x (:183-186)."""
    k, decay, eps = args["num_embeddings"], args["decay"], args["epsilon"]
    emb = state["embedding.weight"].float()
    perm = [0] + list(range(2, x.ndim)) + [1]
    flat = x.detach().float().permute(perm).contiguous().view(-1, emb.shape[1])
    dist = (flat**2).sum(dim=1, keepdim=True) + (emb.t() ** 2).sum(dim=0, keepdim=True) - 2 * torch.mm(flat, emb.t())
    idx = torch.max(-dist, dim=1)[1]
    enc = F.one_hot(idx, k).float()
    shape = list(x.shape)
    del shape[1]
    idx = idx.view(shape)
    back = [0, x.ndim - 1] + list(range(1, x.ndim - 1))
    q = F.embedding(idx, emb).permute(back).contiguous()
    enc_sum, dw = enc.sum(0), torch.mm(enc.t(), flat)                                   # :168-169
    cluster = state["ema_cluster_size"].float() * decay + enc_sum * (1 - decay)           # :174
    n = cluster.sum()                                                                     # :177
    weights = (cluster + eps) / (n + k * eps) * n                                         # :178
    ema_w = state["ema_w"].float() * decay + dw * (1 - decay)                             # :179
    new_state = {"embedding.weight": ema_w / weights.unsqueeze(1), "ema_cluster_size": cluster, "ema_w": ema_w}  # :180
    loss = args["commitment_cost"] * F.mse_loss(q.detach(), x)                            # :183
    return x + (q - x).detach(), loss, idx, new_state                                     # :186


# --------------------------------------------------------------------------------------------------------------------
# Schedulers (networks/schedulers/{scheduler,ddim,ddpm}.py). Tables are fp32 CPU tensors like the reference's.
# --------------------------------------------------------------------------------------------------------------------


def noise_schedule(name, num_train_timesteps, **kw):
    """NoiseSchedules, schedulers/scheduler.py:43-110 -> (betas, alphas, alphas_cumprod)."""
    t = num_train_timesteps
    if name == "linear_beta":
        betas = torch.linspace(kw.get("beta_start", 1e-4), kw.get("beta_end", 2e-2), t, dtype=torch.float32)
    elif name == "scaled_linear_beta":
        betas = torch.linspace(kw.get("beta_start", 1e-4) ** 0.5, kw.get("beta_end", 2e-2) ** 0.5, t,
                               dtype=torch.float32) ** 2
    elif name == "sigmoid_beta":
        b0, b1, r = kw.get("beta_start", 1e-4), kw.get("beta_end", 2e-2), kw.get("sig_range", 6)
        betas = torch.sigmoid(torch.linspace(-r, r, t)) * (b1 - b0) + b0
    elif name == "cosine":
        s = kw.get("s", 8e-3)
        x = torch.linspace(0, t, t + 1)
        ac = torch.cos(((x / t) + s) / (1 + s) * torch.pi * 0.5) ** 2
        ac /= ac[0].item()
        alphas = torch.clip(ac[1:] / ac[:-1], 0.0001, 0.9999)
        return 1.0 - alphas, alphas, ac[:-1]
    else:
        raise ValueError(f"Component '{name}' not found")
    alphas = 1.0 - betas
    return betas, alphas, torch.cumprod(alphas, dim=0)


def inference_timesteps(num_train_timesteps, num_inference_steps, steps_offset=0):
    """set_timesteps, schedulers/ddim.py:123-144 / ddpm.py:111-131."""
    if num_inference_steps > num_train_timesteps:
        raise ValueError("num_inference_steps cannot be larger than num_train_timesteps")
    ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
    return torch.from_numpy(ts) + steps_offset


def _x0_eps(prediction_type, a_t, model_output, sample):
    b_t = 1 - a_t
    if prediction_type == "epsilon":
        return (sample - (b_t**0.5) * model_output) / (a_t**0.5), model_output
    if prediction_type == "sample":
        return model_output, (sample - (a_t**0.5) * model_output) / (b_t**0.5)
    if prediction_type == "v_prediction":
        return (a_t**0.5) * sample - (b_t**0.5) * model_output, (a_t**0.5) * model_output + (b_t**0.5) * sample
    raise ValueError(prediction_type)


def ddim_step(alphas_cumprod, num_train_timesteps, num_inference_steps, model_output, timestep, sample, eta=0.0,
              prediction_type="epsilon", clip_sample=True, clip_values=(-1, 1), final_alpha_cumprod=None, noise=None):
    """DDIMScheduler.step, schedulers/ddim.py:156-237. `noise` replaces the CPU-generator draw of :229-235."""
    final = torch.tensor(1.0) if final_alpha_cumprod is None else final_alpha_cumprod
    prev_t = timestep - num_train_timesteps // num_inference_steps
    a_t = alphas_cumprod[timestep]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else final
    x0, eps = _x0_eps(prediction_type, a_t, model_output, sample)
    if clip_sample:
        x0 = torch.clamp(x0, clip_values[0], clip_values[1])
    variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)  # _get_variance :146-154
    std = eta * variance**0.5
    prev = a_prev**0.5 * x0 + (1 - a_prev - std**2) ** 0.5 * eps
    if eta > 0:
        prev = prev + variance**0.5 * eta * noise
    return prev, x0


def ddpm_step(betas, alphas, alphas_cumprod, model_output, timestep, sample, prediction_type="epsilon",
              variance_type="fixed_small", clip_sample=True, clip_values=(-1, 1), noise=None):
    """DDPMScheduler.step, schedulers/ddpm.py:191-252 with _get_variance :158-189. `noise` = the :244-247 draw."""
    pv = None
    if model_output.shape[1] == sample.shape[1] * 2 and variance_type in ["learned", "learned_range"]:
        model_output, pv = torch.split(model_output, sample.shape[1], dim=1)
    one = torch.tensor(1.0)
    a_t = alphas_cumprod[timestep]
    a_prev = alphas_cumprod[timestep - 1] if timestep > 0 else one
    b_t, b_prev = 1 - a_t, 1 - a_prev
    x0, _ = _x0_eps(prediction_type, a_t, model_output, sample)
    if clip_sample:
        x0 = torch.clamp(x0, clip_values[0], clip_values[1])
    prev = (a_prev**0.5 * betas[timestep]) / b_t * x0 + alphas[timestep] ** 0.5 * b_prev / b_t * sample
    if timestep > 0:
        var = (1 - a_prev) / (1 - a_t) * betas[timestep]
        if variance_type == "fixed_small":
            var = torch.clamp(var, min=1e-20)
        elif variance_type == "fixed_large":
            var = betas[timestep]
        elif variance_type == "learned":
            var = pv
        elif variance_type == "learned_range":
            frac = (pv + 1) / 2
            var = frac * betas[timestep] + (1 - frac) * var
        prev = prev + (var**0.5) * noise
    return prev, x0


def add_noise(alphas_cumprod, original, noise, timesteps):
    """Scheduler.add_noise, schedulers/scheduler.py:169-189 (table cast to the sample dtype first, :182)."""
    ac = alphas_cumprod.to(dtype=original.dtype)
    sa = (ac[timesteps] ** 0.5).reshape(-1, *([1] * (original.ndim - 1)))
    sb = ((1 - ac[timesteps]) ** 0.5).reshape(-1, *([1] * (original.ndim - 1)))
    return sa * original + sb * noise


def get_velocity(alphas_cumprod, sample, noise, timesteps):
    """Scheduler.get_velocity, schedulers/scheduler.py:191-200."""
    ac = alphas_cumprod.to(dtype=sample.dtype)
    sa = (ac[timesteps] ** 0.5).reshape(-1, *([1] * (sample.ndim - 1)))
    sb = ((1 - ac[timesteps]) ** 0.5).reshape(-1, *([1] * (sample.ndim - 1)))
    return sa * noise - sb * sample


def pndm_timesteps(num_train_timesteps, num_inference_steps, skip_prk_steps=False, steps_offset=0, order=4):
    """PNDMScheduler.set_timesteps, schedulers/pndm.py:119-163 -> (prk_timesteps, plms_timesteps, timesteps) as int64 arrays."""
    if num_inference_steps > num_train_timesteps:
        raise ValueError("num_inference_steps cannot be larger than num_train_timesteps")
    ratio = num_train_timesteps // num_inference_steps
    base = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64) + steps_offset
    if skip_prk_steps:
        prk, plms = np.array([]), base[::-1]
    else:
        pairs = np.array(base[-order:]).repeat(2) + np.tile(np.array([0, num_train_timesteps // num_inference_steps // 2]), order)
        prk = (pairs[:-1].repeat(2)[1:-1])[::-1].copy()
        plms = base[:-3][::-1].copy()
    return prk, plms, np.concatenate([prk, plms]).astype(np.int64)


class PNDM:
    """Functional-state restatement of PNDMScheduler.step / step_prk / step_plms / _get_prev_sample (schedulers/pndm.py:165-316).
    Quirk kept: after set_timesteps, `num_inference_steps` is the number of *model evaluations* (pndm.py:158-159), and the
    step ratio used by the step functions is derived from that number."""

    def __init__(self, alphas_cumprod, num_train_timesteps, num_inference_steps, skip_prk_steps=False, set_alpha_to_one=False,
                 prediction_type="epsilon", steps_offset=0):
        self.ac = alphas_cumprod
        self.T = num_train_timesteps
        self.skip = skip_prk_steps
        self.final = torch.tensor(1.0) if set_alpha_to_one else alphas_cumprod[0]
        self.prediction_type = prediction_type
        self.prk, self.plms, ts = pndm_timesteps(num_train_timesteps, num_inference_steps, skip_prk_steps, steps_offset)
        self.timesteps = torch.from_numpy(ts)
        self.n = len(ts)
        self.acc, self.counter, self.cur_sample, self.ets = 0, 0, None, []

    def transfer(self, sample, t, prev_t, e):  # :276-316
        a_t = self.ac[t]
        a_prev = self.ac[prev_t] if prev_t >= 0 else self.final
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.prediction_type == "v_prediction":
            e = (a_t**0.5) * e + (b_t**0.5) * sample
        coeff = (a_prev / a_t) ** (0.5)
        denom = a_t * b_prev ** (0.5) + (a_t * b_t * a_prev) ** (0.5)
        return coeff * sample - (a_prev - a_t) * e / denom

    def step(self, model_output, timestep, sample):
        if self.counter < len(self.prk) and not self.skip:
            return self._prk(model_output, timestep, sample)
        return self._plms(model_output, timestep, sample)

    def _prk(self, mo, t, sample):  # :188-229
        half = 0 if self.counter % 2 else self.T // self.n // 2
        prev_t = t - half
        t = int(self.prk[self.counter // 4 * 4])
        phase = self.counter % 4
        if phase == 0:
            self.acc = self.acc + 1 / 6 * mo
            self.ets.append(mo)
            self.cur_sample = sample
        elif phase in (1, 2):
            self.acc = self.acc + 1 / 3 * mo
        else:
            mo = self.acc + 1 / 6 * mo
            self.acc = 0
        cur = self.cur_sample if self.cur_sample is not None else sample
        out = self.transfer(cur, t, prev_t, mo)
        self.counter += 1
        return out

    def _plms(self, mo, t, sample):  # :231-274
        if not self.skip and len(self.ets) < 3:
            raise ValueError("plms steps need 12 prk iterations first")
        prev_t = t - self.T // self.n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(mo)
        else:
            prev_t = t
            t = t + self.T // self.n
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            mo = (mo + e[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(e) == 2:
            mo = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            mo = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            mo = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        out = self.transfer(sample, t, prev_t, mo)
        self.counter += 1
        return out


def approx_standard_normal_cdf(x):
    """DiffusionInferer._approx_standard_normal_cdf, inferers/inferer.py:258-267 (tanh approximation)."""
    return 0.5 * (1.0 + torch.tanh(torch.sqrt(torch.Tensor([2.0 / math.pi])) * (x + 0.044715 * torch.pow(x, 3))))


def decoder_log_likelihood(inputs, means, log_scales, original_input_range=(0, 255), scaled_input_range=(0, 1)):
    """DiffusionInferer._get_decoder_log_likelihood, inferers/inferer.py:269-321."""
    bin_width = (scaled_input_range[1] - scaled_input_range[0]) / (original_input_range[1] - original_input_range[0])
    centered = inputs - means
    inv_std = torch.exp(-log_scales)
    cdf_hi = approx_standard_normal_cdf(inv_std * (centered + bin_width / 2))
    cdf_lo = approx_standard_normal_cdf(inv_std * (centered - bin_width / 2))
    log_hi = torch.log(cdf_hi.clamp(min=1e-12))
    log_one_minus_lo = torch.log((1.0 - cdf_lo).clamp(min=1e-12))
    log_mid = torch.log((cdf_hi - cdf_lo).clamp(min=1e-12))
    return torch.where(inputs < -0.999, log_hi, torch.where(inputs > 0.999, log_one_minus_lo, log_mid))


def get_likelihood(model_fn, inputs, noise, betas, alphas, alphas_cumprod, timesteps, prediction_type="epsilon",
                   variance_type="fixed_small", clip_sample=True, conditioning=None, mode="crossattn",
                   original_input_range=(0, 255), scaled_input_range=(0, 1)):
    """DiffusionInferer.get_likelihood, inferers/inferer.py:145-256, for fixed variances (the reference's learned-variance branch
    raises at :240). model_fn(x, timesteps, context) -> prediction; `noise` = the single randn_like draw of :189.
    Returns (total_kl (N,), [kl map per timestep])."""
    one = torch.tensor(1.0)
    total = torch.zeros(inputs.shape[0])
    maps = []
    for t in [int(v) for v in timesteps]:
        ts = torch.full(inputs.shape[:1], t).long()
        noisy = add_noise(alphas_cumprod, inputs, noise, ts)
        if mode == "concat":
            mo = model_fn(torch.cat([noisy, conditioning], dim=1), ts, None)
        else:
            mo = model_fn(noisy, ts, conditioning)
        a_t = alphas_cumprod[t]
        a_prev = alphas_cumprod[t - 1] if t > 0 else one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        x0, _ = _x0_eps(prediction_type, a_t, mo, noisy)
        if clip_sample:
            x0 = torch.clamp(x0, -1, 1)
        pred_mean = (a_prev ** (0.5) * betas[t]) / b_t * x0 + alphas[t] ** (0.5) * b_prev / b_t * noisy
        post_mean = a_prev.sqrt() * betas[t] / (1 - a_t) * inputs + alphas[t].sqrt() * (1 - a_prev) / (1 - a_t) * noisy  # ddpm.py:133-156
        var = (1 - a_prev) / (1 - a_t) * betas[t]  # ddpm.py:158-189
        if variance_type == "fixed_small":
            var = torch.clamp(var, min=1e-20)
        elif variance_type == "fixed_large":
            var = betas[t]
        log_post = torch.log(var)
        log_pred = log_post
        if t == 0:
            kl = -decoder_log_likelihood(inputs, pred_mean, 0.5 * log_pred, original_input_range, scaled_input_range)
        else:
            kl = 0.5 * (-1.0 + log_pred - log_post + torch.exp(log_post - log_pred)
                        + ((post_mean - pred_mean) ** 2) * torch.exp(-log_pred))
        total += kl.view(kl.shape[0], -1).mean(axis=1)
        maps.append(kl)
    return total, maps


# --------------------------------------------------------------------------------------------------------------------
# DecoderOnlyTransformer (networks/nets/transformer.py, blocks/transformerblock.py, blocks/selfattention.py) + Ordering
# --------------------------------------------------------------------------------------------------------------------


def sa_block(sd, p, x, heads, context=None, causal=False):
    """SABlock.forward, blocks/selfattention.py:98-147 (non-flash path): q from x, k/v from context or x, scaled scores, optional
    lower-triangular mask, softmax, out_proj."""
    b, t, c = x.shape
    kv = context if context is not None else x
    kt = kv.shape[1]
    hd = c // heads
    q = _lin(sd, f"{p}.to_q", x).view(b, t, heads, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    k = _lin(sd, f"{p}.to_k", kv).view(b, kt, heads, hd).transpose(1, 2)
    v = _lin(sd, f"{p}.to_v", kv).view(b, kt, heads, hd).transpose(1, 2)
    scores = q @ k.transpose(-2, -1)
    if causal:
        scores = scores.masked_fill(torch.tril(torch.ones(t, kt)).view(1, 1, t, kt) == 0, float("-inf"))
    y = (F.softmax(scores, dim=-1) @ v).transpose(1, 2).contiguous().view(b, t, c)
    return _lin(sd, f"{p}.out_proj", y)


def transformer_forward(sd, cfg, tokens, context=None):
    """DecoderOnlyTransformer.forward, nets/transformer.py:98-106: token + absolute position embeddings, pre-norm blocks
    (causal self-attention [+ cross-attention] + GELU MLP, blocks/transformerblock.py:86-91), to_logits.
    cfg: num_tokens, max_seq_len, attn_layers_dim, attn_layers_depth, attn_layers_heads, with_cross_attention."""
    heads, cross = cfg["attn_layers_heads"], cfg.get("with_cross_attention", False)
    b, t = tokens.shape
    x = F.embedding(tokens, sd["token_embeddings.weight"]) + F.embedding(torch.arange(t).repeat(b, 1), sd["position_embeddings.embedding.weight"])
    for i in range(cfg["attn_layers_depth"]):
        p = f"blocks.{i}"
        ln = lambda name, h: F.layer_norm(h, (h.shape[-1],), sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], 1e-5)  # noqa: E731
        x = x + sa_block(sd, f"{p}.attn", ln("norm1", x), heads, causal=True)
        if cross:
            x = x + sa_block(sd, f"{p}.cross_attn", ln("norm2", x), heads, context=context)
        h = ln("norm3", x)
        x = x + _lin(sd, f"{p}.mlp.linear2", F.gelu(_lin(sd, f"{p}.mlp.linear1", h)))
    return _lin(sd, "to_logits", x)


def raster_scan_ordering(dimensions):
    """Ordering('raster_scan', ...) without transformations (utils/ordering.py:113-166): the identity permutation."""
    n = int(np.prod(dimensions[1:]))
    return np.arange(n), np.arange(n)


def ordering_indices(ordering_type, spatial_dims, dimensions, reflected_spatial_dims=(), transpositions_axes=(), rot90_axes=(),
                     transformation_order=("transpose", "rotate_90", "reflect")):
    """utils/ordering.py:52-205 -> (sequence_ordering, revert_sequence_ordering). 'random' is not restated (np.random state)."""
    tmpl = np.arange(int(np.prod(dimensions[1:]))).reshape(*dimensions[1:])
    for tr in transformation_order:
        if tr == "transpose":
            for axes in transpositions_axes:
                tmpl = np.transpose(tmpl, axes=axes)
        elif tr == "rotate_90":
            for axes in rot90_axes:
                tmpl = np.rot90(tmpl, axes=axes)
        elif tr == "reflect":
            for axis, flag in enumerate(reflected_spatial_dims):
                tmpl = np.flip(tmpl, axis=axis) if flag else tmpl
    shape = tmpl.shape
    seq = []
    for r in range(shape[0]):
        cols = range(shape[1]) if (ordering_type == "raster_scan" or r % 2 == 0) else range(shape[1] - 1, -1, -1)
        for c in cols:
            if spatial_dims == 3:
                deps = range(shape[2]) if (ordering_type == "raster_scan" or c % 2 == 0) else range(shape[2] - 1, -1, -1)
                for d in deps:
                    seq.append(tmpl[r, c, d])
            else:
                seq.append(tmpl[r, c])
    order = np.array(seq)
    return order, np.argsort(order)


def transformer_sample_probs(logits_last, temperature, top_k, bos):
    """The sampling head of VQVAETransformerInferer.sample, inferers/inferer.py:1221-1232: temperature, top-k crop, softmax, BOS
    probability zeroed (NOT renormalised)."""
    logits = logits_last / temperature
    if top_k is not None:
        v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
        logits = logits.masked_fill(logits < v[:, [-1]], -float("Inf"))
    probs = F.softmax(logits, dim=-1).clone()
    probs[:, bos] = 0
    return probs


def transformer_likelihood(sd, cfg, latent_idx, ordering, revert, bos, context=None):
    """VQVAETransformerInferer.get_likelihood, inferers/inferer.py:1248-1330, from the VQ indices (B, *spatial) on: log-probability
    of every latent token given its predecessors, re-arranged to the latent grid."""
    shape = latent_idx.shape
    lat = latent_idx.reshape(shape[0], -1)[:, ordering]
    lat = F.pad(lat, (1, 0), "constant", bos).long()
    msl = cfg["max_seq_len"]
    probs = F.softmax(transformer_forward(sd, cfg, lat[:, :msl], context), dim=-1)
    target = lat[:, 1:]
    probs = torch.gather(probs, 2, target[:, :msl].unsqueeze(2)).squeeze(2)
    for i in range(msl, target.shape[1]):
        p = F.softmax(transformer_forward(sd, cfg, lat[:, i + 1 - msl:i + 1], context)[:, -1, :], dim=-1)
        probs = torch.cat((probs, torch.gather(p, 1, target[:, i].unsqueeze(1))), dim=1)
    return torch.log(probs)[:, revert].reshape(shape)


# --------------------------------------------------------------------------------------------------------------------
# Inferer loops (inferers/inferer.py)
# --------------------------------------------------------------------------------------------------------------------


def ddim_sample(sd, cfg, noise, sched, conditioning=None, mode="crossattn", on_step=None):
    """DiffusionInferer.sample, inferers/inferer.py:83-143, with a DDIM scheduler (eta = 0).

    sched: dict(alphas_cumprod, num_train_timesteps, num_inference_steps, timesteps, prediction_type, clip_sample,
    clip_values, final_alpha_cumprod). One shared timestep of shape (1,) per step (:129,133)."""
    image = noise
    for t in sched["timesteps"]:
        ts = torch.Tensor((t,))
        if mode == "concat":
            out = unet_forward(sd, cfg, torch.cat([image, conditioning], dim=1), ts, None)
        else:
            out = unet_forward(sd, cfg, image, ts, conditioning)
        image, _ = ddim_step(sched["alphas_cumprod"], sched["num_train_timesteps"], sched["num_inference_steps"], out,
                             int(t), image, 0.0, sched.get("prediction_type", "epsilon"),
                             sched.get("clip_sample", True), sched.get("clip_values", (-1, 1)),
                             sched.get("final_alpha_cumprod"))
        if on_step is not None:
            on_step(int(t), image)
    return image


def derandomize_zeros(module_or_sd, seed=1234, std=0.05):
    """A fresh reference UNet outputs exact zeros (zero_module'd convs): re-randomise every all-zero
    parameter with N(0, std) from a fixed seed so random-init parity is not vacuous."""
    g = torch.Generator().manual_seed(seed)
    items = module_or_sd.items() if isinstance(module_or_sd, dict) else module_or_sd.named_parameters()
    with torch.no_grad():
        for _, p in items:
            if p.is_floating_point() and p.numel() > 0 and p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * std)


def synthetic_state_dict(shapes, seed=0, dtype=torch.float32):
    """Reference-independent deterministic weights for fixtures too large to commit: every tensor of `shapes`
    (key -> shape, e.g. from a reference model's state_dict) is filled from one seeded CPU generator in sorted-key order:
    matrices / conv kernels ~ N(0, 1/fan_in), norm gains (1-D ``*.weight``) ~ 1 + 0.1 N(0,1), other 1-D ~ 0.05 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        r = torch.randn(shp, generator=g, dtype=torch.float32)
        if len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            r = r / math.sqrt(fan_in)
        elif k.endswith(".weight"):
            r = 1.0 + 0.1 * r
        else:
            r = 0.05 * r
        out[k] = r.to(dtype)
    return out
