"""TEST INFRASTRUCTURE ONLY. Generates tests/golden/*.pt by running the UNMODIFIED reference (/root/reference, via
oracle/monai_stub.py) on CPU fp32 with fixed seeds. Only runnable in the build container; the fixtures are committed so
the GPU box (no /root/reference) can pin both the oracle restatement and the HIP path against real reference outputs.

    python oracle/make_golden.py            # rewrites every fixture

Each fixture: dict(kind, cfg, state_dict, inputs{...}, outputs{...}, meta). Weights: reference default init under
torch.manual_seed(seed), then derandomize_zeros(seed=1234, std=0.05).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402
from restatement import derandomize_zeros, synthetic_state_dict  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


UNET_CASES = {
    # the literal reference test configs, tests/test_diffusion_inferer.py:23-50 (C1a)
    "unet2d_c1a": dict(cfg=dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=[8], norm_num_groups=8,
                                attention_levels=[True], num_res_blocks=1, num_head_channels=8), shape=(2, 1, 8, 8)),
    "unet3d_c1a": dict(cfg=dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=[8], norm_num_groups=8,
                                attention_levels=[True], num_res_blocks=1, num_head_channels=8), shape=(2, 1, 8, 8, 8)),
    # BASELINE.json configs[0] wording (C1b) at reduced resolution
    "unet2d_c1b": dict(cfg=dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(32, 64),
                                attention_levels=(False, True), num_res_blocks=1, num_head_channels=64),
                       shape=(2, 1, 16, 16), synthetic=101),
    # C2 topology (3 levels, 2 res blocks, attention only in the mid block, 1 head) at reduced width / resolution
    "unet3d_c2mini": dict(cfg=dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(32, 64, 64),
                                   attention_levels=(False, False, False), num_res_blocks=2,
                                   num_head_channels=(0, 0, 64), norm_num_groups=32), shape=(1, 1, 16, 16, 16), synthetic=102),
    # conditioned: cross-attention + class embedding + resblock_updown + concat-width groups
    "unet2d_cond": dict(cfg=dict(spatial_dims=2, in_channels=2, out_channels=3, num_channels=(8, 16, 16),
                                 attention_levels=(False, True, True), num_res_blocks=1, norm_num_groups=8,
                                 num_head_channels=4, with_conditioning=True, cross_attention_dim=5,
                                 transformer_num_layers=2, resblock_updown=True, num_class_embeds=4),
                        shape=(2, 2, 8, 8), context=(2, 3, 5), class_labels=[1, 3]),
    "unet3d_cond": dict(cfg=dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(8, 16),
                                 attention_levels=(True, True), num_res_blocks=(1, 2), norm_num_groups=8,
                                 num_head_channels=(8, 4), with_conditioning=True, cross_attention_dim=3,
                                 upcast_attention=True), shape=(2, 1, 8, 8, 8), context=(2, 1, 3)),
}

AEKL_CASES = {
    "aekl2d": dict(cfg=dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(8, 8, 16), latent_channels=4,
                            attention_levels=(False, False, True), num_res_blocks=(1, 1, 2), norm_num_groups=4),
                   shape=(2, 1, 16, 16)),
    "aekl3d_brainlike": dict(cfg=dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4,
                                      num_channels=(8, 16, 16), num_res_blocks=2, norm_num_groups=8,
                                      attention_levels=(False, False, False), with_encoder_nonlocal_attn=False,
                                      with_decoder_nonlocal_attn=False), shape=(1, 1, 16, 16, 16)),
    "aekl3d_convT": dict(cfg=dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(8, 16),
                                  latent_channels=3, attention_levels=(False, False), num_res_blocks=1,
                                  norm_num_groups=8, use_convtranspose=True), shape=(2, 1, 8, 8, 8)),
}

VQVAE_CASES = {
    "vqvae3d": dict(cfg=dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(8, 16), num_res_layers=1,
                             num_res_channels=(8, 16), downsample_parameters=((2, 4, 1, 1),) * 2,
                             upsample_parameters=((2, 4, 1, 1, 0),) * 2, num_embeddings=16, embedding_dim=8),
                    shape=(2, 1, 16, 16, 16)),
    "vqvae2d_odd": dict(cfg=dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(8, 16), num_res_layers=2,
                                 num_res_channels=(4, 8), downsample_parameters=((2, 4, 1, 1), (1, 3, 1, 1)),
                                 upsample_parameters=((1, 3, 1, 1, 0), (2, 3, 1, 1, 1)), num_embeddings=16,
                                 embedding_dim=8, output_act="tanh"), shape=(2, 1, 16, 16)),
}


def bf16_reference_outputs():
    """tests/golden/unet_bf16_ref.pt: the reference's OWN bf16 outputs (model.to(bfloat16) on CPU) for every UNet fixture, so the
    GPU bf16 path can be held to SURVEY.md 8(c)(3): err(ours_bf16) <= 1.5 * err(ref_bf16), both against the fp32 golden."""
    from generative.networks.nets import DiffusionModelUNet

    res = {}
    for name in UNET_CASES:
        fx = torch.load(os.path.join(OUT, name + ".pt"))
        m = DiffusionModelUNet(**fx["cfg"]).eval()
        m.load_state_dict(fx["state_dict"] if fx["state_dict"] is not None else synthetic_state_dict(fx["shapes"], seed=fx["synthetic_seed"]))
        m = m.to(torch.bfloat16)
        i = fx["inputs"]
        ctx = None if i["context"] is None else i["context"].bfloat16()
        with torch.no_grad():
            y = m(i["x"].bfloat16(), i["timesteps"], context=ctx, class_labels=i["class_labels"])
        err = (y.float() - fx["outputs"]["y"]).abs()
        res[name] = dict(y_bf16=y, mean_err=float(err.mean()), max_err=float(err.max()), sigma=float(fx["outputs"]["y"].std()))
        print(name, "reference bf16 vs fp32: mean", res[name]["mean_err"], "max", res[name]["max_err"], "sigma", res[name]["sigma"])
    torch.save(dict(kind="unet_bf16_ref", cases=res), os.path.join(OUT, "unet_bf16_ref.pt"))


def extras():
    """tests/golden/pndm_likelihood.pt: PNDMScheduler tables, a model-free step sequence and a UNet chain; DiffusionInferer.
    get_likelihood totals + per-step maps.  Outputs of the unmodified reference."""
    from generative.inferers import DiffusionInferer
    from generative.networks.nets import DiffusionModelUNet
    from generative.networks.schedulers import DDPMScheduler, PNDMScheduler

    out = dict(kind="pndm_likelihood", tables={}, sequences={}, likelihood={})
    for n in (10, 50, 100):
        for skip in (False, True):
            s = PNDMScheduler(1000, skip_prk_steps=skip)
            s.set_timesteps(n)
            out["tables"][(n, skip)] = dict(prk=torch.as_tensor(s.prk_timesteps.astype("int64")) if len(s.prk_timesteps) else torch.zeros(0, dtype=torch.long),
                                            plms=torch.as_tensor(s.plms_timesteps.copy()), timesteps=s.timesteps.clone(),
                                            num_inference_steps=s.num_inference_steps)
    # model-free sequences: step() fed with fixed pseudo model outputs exercises every branch of the state machine
    shape = (2, 2, 4, 4, 4)
    for sname, kw in [("linear_beta", {}), ("scaled_linear_beta", dict(beta_start=0.0005, beta_end=0.0195))]:
        for pt in ["epsilon", "v_prediction"]:
            for skip in (False, True):
                for one in (False, True):
                    s = PNDMScheduler(1000, schedule=sname, skip_prk_steps=skip, set_alpha_to_one=one, prediction_type=pt, **kw)
                    s.set_timesteps(20)
                    x = _randn(shape, 31)
                    seq = []
                    for k, t in enumerate(s.timesteps):
                        x, _ = s.step(_randn(shape, 100 + k), int(t), x)
                        seq.append(x.clone())
                    out["sequences"][(sname, pt, skip, one)] = dict(kw=kw, x0_seed=31, mo_seed0=100, shape=shape, steps=20, samples=seq)
    # UNet chain (C1a 3D): PNDM sampling through the reference inferer
    torch.manual_seed(0)
    cfg = UNET_CASES["unet3d_c1a"]["cfg"]
    m = DiffusionModelUNet(**cfg).eval()
    derandomize_zeros(m)
    noise = _randn((2, 1, 8, 8, 8), 21)
    pndm = PNDMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195)
    pndm.set_timesteps(10)
    chain = DiffusionInferer(pndm).sample(noise, m, pndm, verbose=False)
    out["chain"] = dict(cfg=cfg, state_dict=m.state_dict(), noise=noise, steps=10, out=chain)
    # get_likelihood (C1a 2D), all fixed-variance / prediction-type combinations; inputs reach past +-0.999 (edge bins)
    torch.manual_seed(0)
    cfg2 = UNET_CASES["unet2d_c1a"]["cfg"]
    m2 = DiffusionModelUNet(**cfg2).eval()
    derandomize_zeros(m2)
    inputs = (_randn((2, 1, 8, 8), 41) * 0.6).clamp(-1, 1)
    out["likelihood"]["cfg"], out["likelihood"]["state_dict"], out["likelihood"]["inputs"] = cfg2, m2.state_dict(), inputs
    out["likelihood"]["noise_seed"] = 123
    out["likelihood"]["cases"] = {}
    for vt in ["fixed_small", "fixed_large"]:
        for pt in ["epsilon", "v_prediction", "sample"]:
            for clip in (True, False):
                d = DDPMScheduler(num_train_timesteps=10, variance_type=vt, prediction_type=pt, clip_sample=clip)
                inf = DiffusionInferer(d)
                torch.manual_seed(123)
                total, maps = inf.get_likelihood(inputs, m2, d, save_intermediates=True, verbose=False)
                out["likelihood"]["cases"][(vt, pt, clip)] = dict(total=total, maps=maps)
    torch.save(out, os.path.join(OUT, "pndm_likelihood.pt"))
    print("pndm tables", {k: int(v["num_inference_steps"]) for k, v in out["tables"].items()})
    print("chain", float(chain.abs().max()), "likelihood", {k: v["total"].tolist() for k, v in list(out["likelihood"]["cases"].items())[:3]})


def controlnet_fixtures():
    """tests/golden/controlnet.pt: ControlNet forwards (2-D with class embedding + attention, 3-D with cross-attention), the
    ControlNet-conditioned inferer (__call__ + a DDIM-4 chain) and its latent variant with a conditioning image that the reference
    resizes with F.interpolate(nearest).  Outputs of the unmodified reference."""
    from generative.inferers import ControlNetDiffusionInferer, ControlNetLatentDiffusionInferer
    from generative.networks.nets import AutoencoderKL, ControlNet, DiffusionModelUNet
    from generative.networks.schedulers import DDIMScheduler

    out = dict(kind="controlnet", forwards={})
    cases = {
        "cn2d": dict(cfg=dict(spatial_dims=2, in_channels=1, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1,
                              norm_num_groups=8, num_head_channels=8, num_class_embeds=3, conditioning_embedding_in_channels=2,
                              conditioning_embedding_num_channels=(8, 16)), x=(2, 1, 8, 8), cond=(2, 2, 16, 16), class_labels=[0, 2]),
        "cn3d_cross": dict(cfg=dict(spatial_dims=3, in_channels=1, num_channels=(8, 8), attention_levels=(True, True), num_res_blocks=(1, 2),
                                    norm_num_groups=8, num_head_channels=4, with_conditioning=True, cross_attention_dim=3,
                                    conditioning_embedding_in_channels=1, conditioning_embedding_num_channels=(8,)),
                           x=(2, 1, 8, 8, 8), cond=(2, 1, 8, 8, 8), context=(2, 2, 3)),
    }
    for name, case in cases.items():
        torch.manual_seed(0)
        m = ControlNet(**case["cfg"]).eval()
        derandomize_zeros(m)
        x, cond = _randn(case["x"], 7), _randn(case["cond"], 9)
        t = torch.tensor([980, 20])
        ctx = _randn(case["context"], 8) if "context" in case else None
        cl = torch.tensor(case["class_labels"]) if "class_labels" in case else None
        with torch.no_grad():
            down, mid = m(x, t, cond, conditioning_scale=0.7, context=ctx, class_labels=cl)
        out["forwards"][name] = dict(cfg=case["cfg"], state_dict=m.state_dict(), x=x, timesteps=t, cond=cond, context=ctx, class_labels=cl,
                                     scale=0.7, down=[d.clone() for d in down], mid=mid.clone())
        print(name, len(down), float(mid.abs().max()))
    # conditioned sampling: 2-D UNet + ControlNet
    ucfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1,
                norm_num_groups=8, num_head_channels=8)
    ccfg = dict(spatial_dims=2, in_channels=1, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1, norm_num_groups=8,
                num_head_channels=8, conditioning_embedding_in_channels=1, conditioning_embedding_num_channels=(8, 16))
    torch.manual_seed(0)
    unet = DiffusionModelUNet(**ucfg).eval()
    derandomize_zeros(unet)
    torch.manual_seed(1)
    cn = ControlNet(**ccfg).eval()
    derandomize_zeros(cn)
    noise, cond = _randn((2, 1, 8, 8), 21), _randn((2, 1, 16, 16), 23)
    sch = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sch.set_timesteps(4)
    inf = ControlNetDiffusionInferer(sch)
    chain = inf.sample(noise, unet, cn, cond, sch, verbose=False)
    xin, ts = _randn((2, 1, 8, 8), 22), torch.tensor([700, 30])
    pred = inf(inputs=xin, diffusion_model=unet, controlnet=cn, noise=noise, timesteps=ts, cn_cond=cond)
    out["inferer"] = dict(unet_cfg=ucfg, unet_sd=unet.state_dict(), cn_cfg=ccfg, cn_sd=cn.state_dict(), noise=noise, cond=cond, steps=4,
                          chain=chain, call_inputs=xin, call_timesteps=ts, call_prediction=pred)
    # latent variant: AutoencoderKL 16x16 -> 4x4 latent of 4 channels; cond given at image size 20x20 is resized to 8x8 = 2 x latent
    acfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(8, 8, 16), latent_channels=4, attention_levels=(False, False, False),
                num_res_blocks=1, norm_num_groups=4, with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)
    lucfg = dict(ucfg, in_channels=4, out_channels=4)
    lccfg = dict(ccfg, in_channels=4, conditioning_embedding_num_channels=(8,))
    torch.manual_seed(2)
    ae = AutoencoderKL(**acfg).eval()
    torch.manual_seed(3)
    lunet = DiffusionModelUNet(**lucfg).eval()
    derandomize_zeros(lunet)
    torch.manual_seed(4)
    lcn = ControlNet(**lccfg).eval()
    derandomize_zeros(lcn)
    lnoise, lcond = _randn((2, 4, 4, 4), 31), _randn((2, 1, 10, 6), 32)
    linf = ControlNetLatentDiffusionInferer(sch, scale_factor=0.9)
    limg = linf.sample(lnoise, ae, lunet, lcn, lcond, sch, verbose=False)
    out["latent"] = dict(ae_cfg=acfg, ae_sd=ae.state_dict(), unet_cfg=lucfg, unet_sd=lunet.state_dict(), cn_cfg=lccfg, cn_sd=lcn.state_dict(),
                         noise=lnoise, cond=lcond, steps=4, scale_factor=0.9, image=limg)
    torch.save(out, os.path.join(OUT, "controlnet.pt"))
    print("inferer chain", float(chain.abs().max()), "latent image", tuple(limg.shape), float(limg.abs().max()))


def transformer_fixtures():
    """tests/golden/transformer.pt: DecoderOnlyTransformer forwards (with / without cross-attention), Ordering permutations,
    VQVAETransformerInferer __call__ / get_likelihood / greedy (top_k = 1) sample with a small 2-D VQVAE."""
    from generative.inferers import VQVAETransformerInferer
    from generative.networks.nets import VQVAE, DecoderOnlyTransformer
    from generative.utils.ordering import Ordering

    out = dict(kind="transformer", forwards={}, orderings={})
    for name, cfg, ctx in [("plain", dict(num_tokens=17, max_seq_len=24, attn_layers_dim=32, attn_layers_depth=2, attn_layers_heads=4), None),
                           ("cross", dict(num_tokens=10, max_seq_len=12, attn_layers_dim=16, attn_layers_depth=1, attn_layers_heads=2,
                                          with_cross_attention=True), (2, 3, 16))]:
        torch.manual_seed(0)
        m = DecoderOnlyTransformer(**cfg).eval()
        tok = torch.randint(0, cfg["num_tokens"], (2, cfg["max_seq_len"] - 3), generator=torch.Generator().manual_seed(5))
        context = _randn(ctx, 6) if ctx else None
        with torch.no_grad():
            y = m(tok, context=context)
        out["forwards"][name] = dict(cfg=cfg, state_dict=m.state_dict(), tokens=tok, context=context, logits=y)
        print(name, tuple(y.shape), float(y.abs().max()))
    for key, kw in {"raster2d": dict(ordering_type="raster_scan", spatial_dims=2, dimensions=(1, 4, 6)),
                    "scurve3d": dict(ordering_type="s_curve", spatial_dims=3, dimensions=(1, 3, 4, 2)),
                    "scurve2d_tf": dict(ordering_type="s_curve", spatial_dims=2, dimensions=(1, 4, 4), reflected_spatial_dims=(True, False),
                                        transpositions_axes=((1, 0),), rot90_axes=((0, 1),))}.items():
        o = Ordering(**kw)
        out["orderings"][key] = dict(kw=kw, order=torch.as_tensor(o.get_sequence_ordering().copy()),
                                     revert=torch.as_tensor(o.get_revert_sequence_ordering().copy()))
    vcfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(8, 8), num_res_layers=1, num_res_channels=(8, 8),
                downsample_parameters=((2, 4, 1, 1),) * 2, upsample_parameters=((2, 4, 1, 1, 0),) * 2, num_embeddings=16, embedding_dim=8)
    tcfg = dict(num_tokens=17, max_seq_len=16, attn_layers_dim=32, attn_layers_depth=2, attn_layers_heads=4)
    torch.manual_seed(1)
    vq = VQVAE(**vcfg).eval()
    torch.manual_seed(2)
    tr = DecoderOnlyTransformer(**tcfg).eval()
    x = _randn((2, 1, 16, 16), 7)  # -> 4 x 4 latent = 16 tokens = max_seq_len
    order = Ordering(ordering_type="s_curve", spatial_dims=2, dimensions=(1, 4, 4))
    inf = VQVAETransformerInferer()
    with torch.no_grad():
        pred, target, lsd = inf(x, vq, tr, order, return_latent=True)
        lik = inf.get_likelihood(x, vq, tr, order)
        start = torch.full((2, 1), 16)
        greedy = inf.sample((4, 4), start, vq, tr, order, top_k=1, verbose=False)
        # teacher-forced sampling-head probabilities along the greedy path for temperature / top-k variants
        idx = vq.index_quantize(x).reshape(2, -1)[:, order.get_sequence_ordering()]
        seq = torch.cat([start, idx], 1).long()
        logits = tr(seq[:, :16])
    out["inferer"] = dict(vq_cfg=vcfg, vq_sd=vq.state_dict(), tr_cfg=tcfg, tr_sd=tr.state_dict(), x=x, ordering=dict(ordering_type="s_curve", spatial_dims=2, dimensions=(1, 4, 4)),
                          prediction=pred, target=target, latent_spatial_dim=lsd, likelihood=lik, greedy_image=greedy, seq=seq, logits=logits)
    # a longer-than-context case for get_likelihood's sliding window: 8 x 8 input -> 2 x 2 ... use max_seq_len 3 < 4 tokens
    tcfg2 = dict(num_tokens=17, max_seq_len=3, attn_layers_dim=16, attn_layers_depth=1, attn_layers_heads=2)
    torch.manual_seed(3)
    tr2 = DecoderOnlyTransformer(**tcfg2).eval()
    x2 = _randn((2, 1, 8, 8), 8)
    order2 = Ordering(ordering_type="raster_scan", spatial_dims=2, dimensions=(1, 2, 2))
    with torch.no_grad():
        lik2 = inf.get_likelihood(x2, vq, tr2, order2)
    out["window"] = dict(tr_cfg=tcfg2, tr_sd=tr2.state_dict(), x=x2, likelihood=lik2)
    torch.save(out, os.path.join(OUT, "transformer.pt"))
    print("likelihood", tuple(lik.shape), float(lik.min()), "greedy", tuple(greedy.shape))


def grid_cases():
    """Constructor-argument grid in the spirit of the reference's own shape tests (tests/test_diffusion_model_unet.py:23-232,
    tests/test_autoencoderkl.py:22-160, tests/test_vqvae.py:22-90): per-level res-block tuples, resblock_updown, attention placements and
    head widths, cross-attention with several transformer layers, class embeddings, encoder / decoder non-local attention switches,
    ConvTranspose up-sampling, scalar-vs-tuple VQ-VAE parameters -- in 2-D and 3-D."""
    unet = []
    for sd in (2, 3):
        base = dict(spatial_dims=sd, in_channels=1, out_channels=1, num_channels=(8, 8, 8), norm_num_groups=8)
        unet += [
            dict(base, num_res_blocks=1, attention_levels=(False, False, False)),
            dict(base, num_res_blocks=(1, 1, 2), attention_levels=(False, False, False)),
            dict(base, num_res_blocks=1, attention_levels=(False, False, False), resblock_updown=True),
            dict(base, num_res_blocks=1, attention_levels=(False, False, True), num_head_channels=8),
            dict(base, num_res_blocks=1, attention_levels=(False, False, True), num_head_channels=4, resblock_updown=True),
            dict(base, num_res_blocks=1, attention_levels=(False, True, True), num_head_channels=(0, 2, 4)),
            dict(base, num_res_blocks=1, attention_levels=(False, False, True), num_head_channels=4, with_conditioning=True,
                 cross_attention_dim=3, transformer_num_layers=2),
            dict(base, in_channels=2, out_channels=3, num_res_blocks=(2, 1, 1), attention_levels=(True, False, True), num_head_channels=8,
                 num_class_embeds=5, upcast_attention=True),
        ]
    aekl = []
    for sd in (2, 3):
        base = dict(spatial_dims=sd, in_channels=1, out_channels=1, num_channels=(4, 4, 4), latent_channels=4, norm_num_groups=4)
        aekl += [
            dict(base, attention_levels=(False, False, False), num_res_blocks=1),
            dict(base, attention_levels=(False, False, False), num_res_blocks=(1, 1, 2)),
            dict(base, attention_levels=(False, False, True), num_res_blocks=1),
            dict(base, attention_levels=(False, False, False), num_res_blocks=1, with_encoder_nonlocal_attn=False),
            dict(base, attention_levels=(False, True, False), num_res_blocks=1, with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False,
                 use_convtranspose=True),
        ]
    vq = []
    for sd in (2, 3):
        base = dict(spatial_dims=sd, in_channels=1, out_channels=1, num_channels=(4, 4), num_res_layers=1, num_embeddings=8, embedding_dim=8)
        vq += [
            dict(base, num_res_channels=(4, 4), downsample_parameters=((2, 4, 1, 1),) * 2, upsample_parameters=((2, 4, 1, 1, 0),) * 2),
            dict(base, num_res_channels=4, downsample_parameters=(2, 4, 1, 1), upsample_parameters=(2, 4, 1, 1, 0)),
            dict(base, num_res_layers=2, num_res_channels=(4, 8), downsample_parameters=((2, 4, 1, 1), (1, 3, 1, 1)),
                 upsample_parameters=((1, 3, 1, 1, 0), (2, 4, 1, 1, 0)), act="LEAKYRELU", output_act="sigmoid"),
        ]
    return unet, aekl, vq


def grid_fixture():
    """tests/golden/config_grid.pt: reference outputs over grid_cases(); weights are synthetic_state_dict(shapes, seed) on both sides."""
    from generative.networks.nets import VQVAE, AutoencoderKL, DiffusionModelUNet

    unet, aekl, vq = grid_cases()
    out = dict(kind="config_grid", unet=[], aekl=[], vqvae=[])
    for i, cfg in enumerate(unet):
        m = DiffusionModelUNet(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synthetic_state_dict(shapes, seed=300 + i))
        sp = (16,) * cfg["spatial_dims"]
        x = _randn((2, cfg["in_channels"], *sp), 7)
        ctx = _randn((2, 2, cfg["cross_attention_dim"]), 8) if cfg.get("with_conditioning") else None
        cl = torch.tensor([1, 4]) if cfg.get("num_class_embeds") else None
        t = torch.tensor([600, 30])
        with torch.no_grad():
            y = m(x, t, context=ctx, class_labels=cl)
        out["unet"].append(dict(cfg=cfg, shapes=shapes, seed=300 + i, x=x, timesteps=t, context=ctx, class_labels=cl, y=y))
    for i, cfg in enumerate(aekl):
        m = AutoencoderKL(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synthetic_state_dict(shapes, seed=400 + i))
        x = _randn((1, 1, *((16,) * cfg["spatial_dims"])), 7)
        with torch.no_grad():
            mu, sigma = m.encode(x)
            rec = m.decode(mu)
        out["aekl"].append(dict(cfg=cfg, shapes=shapes, seed=400 + i, x=x, z_mu=mu, z_sigma=sigma, reconstruction=rec))
    for i, cfg in enumerate(vq):
        m = VQVAE(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synthetic_state_dict(shapes, seed=500 + i))
        x = _randn((1, 1, *((8,) * cfg["spatial_dims"])), 7)
        with torch.no_grad():
            z = m.encode(x)
            idx = m.index_quantize(x)
            rec = m.decode_samples(idx)
        out["vqvae"].append(dict(cfg=cfg, shapes=shapes, seed=500 + i, x=x, z=z, indices=idx, reconstruction=rec))
    torch.save(out, os.path.join(OUT, "config_grid.pt"))
    print("grid", len(out["unet"]), len(out["aekl"]), len(out["vqvae"]))


def spade_fixtures():
    """tests/golden/spade.pt: SPADE block, SPADEDiffusionModelUNet forwards (2-D with attention + a coarser segmentation, 3-D with
    cross-attention), SPADEAutoencoderKL forward / decode, and a SPADE latent-diffusion chain through LatentDiffusionInferer.sample
    (seg handed to both networks).  Outputs of the unmodified reference."""
    from generative.inferers import LatentDiffusionInferer
    from generative.networks.blocks.spade_norm import SPADE
    from generative.networks.nets import SPADEAutoencoderKL, SPADEDiffusionModelUNet
    from generative.networks.schedulers import DDIMScheduler

    out = dict(kind="spade", blocks={}, unets={}, aekls={})
    for name, kw, xs, ss in [("inst2d", dict(label_nc=3, norm_nc=8, spatial_dims=2, hidden_channels=16), (2, 8, 6, 10), (2, 3, 12, 20)),
                             ("group3d", dict(label_nc=2, norm_nc=16, spatial_dims=3, hidden_channels=8, norm="GROUP",
                                              norm_params={"num_groups": 4, "eps": 1e-6, "affine": True}), (1, 16, 4, 6, 8), (1, 2, 4, 6, 8))]:
        torch.manual_seed(0)
        m = SPADE(**kw).eval()
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(torch.randn(p_.shape, generator=torch.Generator().manual_seed(p_.numel())) * 0.05)
        x, seg = _randn(xs, 41), _randn(ss, 42)
        with torch.no_grad():
            y = m(x, seg)
        out["blocks"][name] = dict(kwargs=kw, state_dict=m.state_dict(), x=x, seg=seg, y=y)
        print("spade block", name, float(y.abs().max()))
    ucases = {
        "spade_unet2d": dict(cfg=dict(spatial_dims=2, in_channels=1, out_channels=1, label_nc=3, num_channels=(8, 16), attention_levels=(False, True),
                                      num_res_blocks=1, norm_num_groups=8, num_head_channels=8, spade_intermediate_channels=16),
                             x=(2, 1, 16, 16), seg=(2, 3, 8, 8)),
        "spade_unet3d_cross": dict(cfg=dict(spatial_dims=3, in_channels=1, out_channels=2, label_nc=2, num_channels=(8, 8), attention_levels=(True, True),
                                            num_res_blocks=(1, 2), norm_num_groups=8, num_head_channels=4, with_conditioning=True,
                                            cross_attention_dim=3, resblock_updown=True, spade_intermediate_channels=8),
                                   x=(2, 1, 8, 8, 8), seg=(2, 2, 8, 8, 8), context=(2, 2, 3)),
    }
    for name, case in ucases.items():
        torch.manual_seed(0)
        m = SPADEDiffusionModelUNet(**case["cfg"]).eval()
        derandomize_zeros(m)
        x, seg = _randn(case["x"], 7), _randn(case["seg"], 9)
        t = torch.tensor([980, 20])
        ctx = _randn(case["context"], 8) if "context" in case else None
        with torch.no_grad():
            y = m(x, t, seg, context=ctx)
        out["unets"][name] = dict(cfg=case["cfg"], state_dict=m.state_dict(), x=x, timesteps=t, seg=seg, context=ctx, y=y)
        print(name, tuple(y.shape), float(y.abs().max()))
    acases = {
        "spade_aekl2d": dict(cfg=dict(spatial_dims=2, label_nc=3, in_channels=1, out_channels=1, num_channels=(8, 8, 16), latent_channels=4,
                                      attention_levels=(False, False, True), num_res_blocks=(1, 1, 2), norm_num_groups=4,
                                      spade_intermediate_channels=16), x=(2, 1, 16, 16), seg=(2, 3, 16, 16)),
        "spade_aekl3d": dict(cfg=dict(spatial_dims=3, label_nc=2, in_channels=1, out_channels=1, num_channels=(8, 16), latent_channels=4,
                                      attention_levels=(False, False), num_res_blocks=1, norm_num_groups=8, with_encoder_nonlocal_attn=False,
                                      with_decoder_nonlocal_attn=False, spade_intermediate_channels=8), x=(1, 1, 8, 8, 8), seg=(1, 2, 4, 4, 4)),
    }
    for name, case in acases.items():
        torch.manual_seed(0)
        m = SPADEAutoencoderKL(**case["cfg"]).eval()
        x, seg = _randn(case["x"], 11), _randn(case["seg"], 12)
        with torch.no_grad():
            z_mu, z_sigma = m.encode(x)
            dec = m.decode(z_mu, seg)
        out["aekls"][name] = dict(cfg=case["cfg"], state_dict=m.state_dict(), x=x, seg=seg, z_mu=z_mu, z_sigma=z_sigma, decoded=dec)
        print(name, tuple(dec.shape), float(dec.abs().max()))
    # SPADE latent diffusion: 2-D SPADE AE (16x16 -> 4x4x4) + SPADE UNet on the latent, DDIM-4, seg at image resolution
    acfg = acases["spade_aekl2d"]["cfg"]
    ucfg = dict(spatial_dims=2, in_channels=4, out_channels=4, label_nc=3, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1,
                norm_num_groups=8, num_head_channels=8, spade_intermediate_channels=16)
    torch.manual_seed(2)
    ae = SPADEAutoencoderKL(**acfg).eval()
    torch.manual_seed(3)
    unet = SPADEDiffusionModelUNet(**ucfg).eval()
    derandomize_zeros(unet)
    noise, seg = _randn((2, 4, 4, 4), 51), _randn((2, 3, 16, 16), 52)
    sch = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sch.set_timesteps(4)
    inf = LatentDiffusionInferer(sch, scale_factor=0.8)
    img = inf.sample(noise, ae, unet, sch, verbose=False, seg=seg)
    xin, ts = _randn((2, 1, 16, 16), 53), torch.tensor([700, 30])
    torch.manual_seed(9)  # the AE's reparametrisation draw inside __call__ is not compared: only the chain is
    out["latent"] = dict(ae_cfg=acfg, ae_sd=ae.state_dict(), unet_cfg=ucfg, unet_sd=unet.state_dict(), noise=noise, seg=seg, steps=4,
                         scale_factor=0.8, image=img)
    torch.save(out, os.path.join(OUT, "spade.pt"))
    print("spade latent chain", tuple(img.shape), float(img.abs().max()))


def main():
    g = load_reference()
    if g is None:
        raise SystemExit("reference tree not found; golden fixtures can only be made in the build container")
    if "--grid-only" in sys.argv:
        grid_fixture()
        return
    if "--transformer-only" in sys.argv:
        transformer_fixtures()
        return
    if "--controlnet-only" in sys.argv:
        controlnet_fixtures()
        return
    if "--spade-only" in sys.argv:
        spade_fixtures()
        return
    if "--extras-only" in sys.argv:  # PNDM + get_likelihood fixtures only
        extras()
        return
    if "--bf16-only" in sys.argv:  # adds the bf16 reference outputs without rewriting the fp32 fixtures
        bf16_reference_outputs()
        return
    if g is None:
        raise SystemExit("reference tree not found; golden fixtures can only be made in the build container")
    from generative.inferers import DiffusionInferer
    from generative.networks.nets import VQVAE, AutoencoderKL, DiffusionModelUNet
    from generative.networks.schedulers import DDIMScheduler, DDPMScheduler

    os.makedirs(OUT, exist_ok=True)
    for name, case in UNET_CASES.items():
        torch.manual_seed(0)
        m = DiffusionModelUNet(**case["cfg"]).eval()
        derandomize_zeros(m)
        shapes = None
        if case.get("synthetic"):  # weights too large to commit: regenerate from (shapes, seed) on both sides
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            m.load_state_dict(synthetic_state_dict(shapes, seed=case["synthetic"]))
        x = _randn(case["shape"], 7)
        n = case["shape"][0]
        t = torch.tensor([980, 20][:n]) if n > 1 else torch.tensor([500])
        ctx = _randn(case["context"], 8) if "context" in case else None
        cl = torch.tensor(case["class_labels"]) if "class_labels" in case else None
        with torch.no_grad():
            y = m(x, t, context=ctx, class_labels=cl)
        torch.save(dict(kind="unet", cfg=case["cfg"], state_dict=None if shapes else m.state_dict(), shapes=shapes,
                        synthetic_seed=case.get("synthetic"),
                        inputs=dict(x=x, timesteps=t, context=ctx, class_labels=cl), outputs=dict(y=y)),
                   os.path.join(OUT, name + ".pt"))
        print(name, tuple(y.shape), float(y.abs().max()))

    for name, case in AEKL_CASES.items():
        torch.manual_seed(0)
        m = AutoencoderKL(**case["cfg"]).eval()
        x = _randn(case["shape"], 7)
        with torch.no_grad():
            mu, sigma = m.encode(x)
            rec = m.decode(mu)
        torch.save(dict(kind="aekl", cfg=case["cfg"], state_dict=m.state_dict(), inputs=dict(x=x),
                        outputs=dict(z_mu=mu, z_sigma=sigma, reconstruction=rec)), os.path.join(OUT, name + ".pt"))
        print(name, tuple(mu.shape), tuple(rec.shape))

    for name, case in VQVAE_CASES.items():
        torch.manual_seed(0)
        m = VQVAE(**case["cfg"]).eval()
        x = _randn(case["shape"], 7)
        with torch.no_grad():
            z = m.encode(x)
            idx = m.index_quantize(x)
            q, loss = m.quantize(z)
            rec = m.decode(q)
        torch.save(dict(kind="vqvae", cfg=case["cfg"], state_dict=m.state_dict(), inputs=dict(x=x),
                        outputs=dict(z=z, indices=idx, quantized=q, loss=loss, reconstruction=rec)),
                   os.path.join(OUT, name + ".pt"))
        print(name, tuple(z.shape), tuple(idx.shape))

    # scheduler known-answer vectors: tables + single steps on fixed tensors (fp32, bit-exact targets)
    sched = {}
    mo, xs = _randn((2, 2, 4, 4, 4), 11), _randn((2, 2, 4, 4, 4), 12)
    for sname, kw in [("linear_beta", {}), ("scaled_linear_beta", dict(beta_start=0.0005, beta_end=0.0195)),
                      ("sigmoid_beta", {}), ("cosine", {})]:
        ddim = DDIMScheduler(1000, schedule=sname, clip_sample=False, **kw)
        ddim.set_timesteps(50)
        entry = dict(kw=kw, betas=ddim.betas.clone(), alphas=ddim.alphas.clone(),
                     alphas_cumprod=ddim.alphas_cumprod.clone(), timesteps50=ddim.timesteps.clone(), ddim={}, ddpm={})
        for pt in ["epsilon", "sample", "v_prediction"]:
            for clip in [False, True]:
                ddim.prediction_type, ddim.clip_sample = pt, clip
                for t in [980, 500, 20]:
                    entry["ddim"][(pt, clip, t, 0.0)] = ddim.step(mo, t, xs)
                gen = torch.Generator().manual_seed(5)
                entry["ddim"][(pt, clip, 500, 0.5)] = ddim.step(mo, 500, xs, eta=0.5, generator=gen)
        ddpm = DDPMScheduler(1000, schedule=sname, **kw)
        for pt in ["epsilon", "sample", "v_prediction"]:
            for vt in ["fixed_small", "fixed_large"]:
                ddpm.prediction_type, ddpm.variance_type = pt, vt
                for t in [999, 500, 1]:
                    gen = torch.Generator().manual_seed(5)
                    entry["ddpm"][(pt, vt, t)] = ddpm.step(mo, t, xs, generator=gen)
        ts = torch.tensor([999, 3])
        entry["add_noise"] = ddpm.add_noise(xs, mo, ts)
        entry["get_velocity"] = DDPMScheduler(1000, schedule=sname, **kw).get_velocity(xs, mo, ts)
        sched[sname] = entry
    # learned variance (model predicts 2*C channels)
    mo2 = _randn((2, 4, 4, 4, 4), 13)
    for vt in ["learned", "learned_range"]:
        d = DDPMScheduler(1000, variance_type=vt)
        gen = torch.Generator().manual_seed(5)
        sched["linear_beta"]["ddpm"][("epsilon", vt, 500)] = d.step(mo2, 500, xs, generator=gen)
    torch.save(dict(kind="schedulers", model_output=mo, model_output2=mo2, sample=xs, noise_seed=5, tables=sched),
               os.path.join(OUT, "schedulers.pt"))

    # a full (free-running) DDIM-10 chain of the C1a 3D model, clip_sample=False, and a DDPM-10 chain with seeded noise
    torch.manual_seed(0)
    cfg = UNET_CASES["unet3d_c1a"]["cfg"]
    m = DiffusionModelUNet(**cfg).eval()
    derandomize_zeros(m)
    noise = _randn((2, 1, 8, 8, 8), 21)
    ddim = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    ddim.set_timesteps(10)
    inf = DiffusionInferer(ddim)
    out, inter = inf.sample(noise, m, ddim, save_intermediates=True, intermediate_steps=100, verbose=False)
    ddpm = DDPMScheduler(num_train_timesteps=10)
    ddpm.set_timesteps(10)
    torch.manual_seed(99)  # DDPM draws its noise from the global CPU generator (ddpm.py:244-247)
    out_p, inter_p = DiffusionInferer(ddpm).sample(noise, m, ddpm, save_intermediates=True, intermediate_steps=1,
                                                   verbose=False)
    xin = _randn((2, 1, 8, 8, 8), 22)
    ts = torch.tensor([7, 2])
    pred = DiffusionInferer(ddpm)(inputs=xin, diffusion_model=m, noise=noise, timesteps=ts)
    torch.save(dict(kind="chain", cfg=cfg, state_dict=m.state_dict(), noise=noise, ddim_out=out, ddim_inter=inter,
                    ddpm_out=out_p, ddpm_inter=inter_p, ddpm_global_seed=99, call_inputs=xin, call_timesteps=ts,
                    call_prediction=pred), os.path.join(OUT, "chain_c1a3d.pt"))
    print("chains", float(out.abs().max()), float(out_p.abs().max()), len(inter), len(inter_p))


if __name__ == "__main__":
    main()
