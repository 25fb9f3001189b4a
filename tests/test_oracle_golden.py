"""CPU: pins the oracle restatement (oracle/restatement.py) against golden outputs of the UNMODIFIED reference
(tests/golden/*.pt, made by oracle/make_golden.py in the build container). Tolerance 2e-5 abs on O(1) outputs = the
reference's own fp32 op-order noise floor (fp32-vs-fp64 1.2e-5); scheduler arithmetic is bit-exact."""
import pytest
import torch

import restatement as R
from _util import assert_close, load_fixture

UNETS = ["unet2d_c1a", "unet3d_c1a", "unet2d_c1b", "unet3d_c2mini", "unet2d_cond", "unet3d_cond"]


@pytest.mark.parametrize("name", UNETS)
def test_unet_forward_matches_reference(name):
    fx = load_fixture(name)
    i = fx["inputs"]
    with torch.no_grad():
        y = R.unet_forward(fx["state_dict"], fx["cfg"], i["x"], i["timesteps"], i["context"], i["class_labels"])
    assert_close(y, fx["outputs"]["y"], atol=2e-5, what=name)


@pytest.mark.parametrize("name", ["aekl2d", "aekl3d_brainlike", "aekl3d_convT"])
def test_aekl_matches_reference(name):
    fx = load_fixture(name)
    sd, cfg, x = fx["state_dict"], fx["cfg"], fx["inputs"]["x"]
    with torch.no_grad():
        mu, sigma = R.aekl_encode(sd, cfg, x)
        rec = R.aekl_decode(sd, cfg, fx["outputs"]["z_mu"])
    assert_close(mu, fx["outputs"]["z_mu"], 2e-5, what="z_mu")
    assert_close(sigma, fx["outputs"]["z_sigma"], 2e-5, what="z_sigma")
    assert_close(rec, fx["outputs"]["reconstruction"], 2e-5, what="reconstruction")


@pytest.mark.parametrize("name", ["vqvae3d", "vqvae2d_odd"])
def test_vqvae_matches_reference(name):
    fx = load_fixture(name)
    sd, cfg, x, o = fx["state_dict"], fx["cfg"], fx["inputs"]["x"], fx["outputs"]
    with torch.no_grad():
        z = R.vqvae_encode(sd, cfg, x)
        idx, _ = R.vq_index_quantize(sd, o["z"])
        q, loss = R.vq_quantize(sd, cfg, o["z"])
        rec = R.vqvae_decode(sd, cfg, o["quantized"])
    assert_close(z, o["z"], 2e-5, what="z")
    assert torch.equal(idx, o["indices"])  # integer work: bit-exact
    assert_close(q, o["quantized"], 1e-6, what="quantized")
    assert_close(loss, o["loss"], 1e-6, what="loss")
    assert_close(rec, o["reconstruction"], 2e-5, what="reconstruction")
    assert_close(R.vqvae_decode(sd, cfg, R.vq_embed(sd, o["indices"])), o["reconstruction"], 2e-5, what="decode_samples")


def test_vq_ema_training_forward_matches_reference():
    """The restated EMA codebook update against two consecutive train() forwards of the unmodified reference EMAQuantizer
    (tests/golden/vq_ema.pt, oracle/make_golden_vq_ema.py): outputs, updated buffers, input gradient."""
    fx = load_fixture("vq_ema")
    for name, case in fx["cases"].items():
        state = {k: v.clone() for k, v in case["init"].items()}
        for s, st in enumerate(case["steps"]):
            x = st["x"].clone().requires_grad_(True)
            q, loss, idx, state = R.vq_ema_forward(state, case["args"], x)
            ((q * st["gq"]).sum() + 3.0 * loss).backward()
            assert torch.equal(idx, st["indices"]), (name, s)
            assert_close(q, st["quantized"], 1e-6, what=f"{name} step {s} quantized")
            assert_close(loss, st["loss"], 1e-6, what=f"{name} step {s} loss")
            assert_close(x.grad, st["dx"], 1e-6, what=f"{name} step {s} dx")
            for k in ("embedding.weight", "ema_cluster_size", "ema_w"):
                assert_close(state[k], st["state"][k], 1e-6, what=f"{name} step {s} {k}")


def _eq(a, b):
    return torch.allclose(a, b, rtol=0, atol=0, equal_nan=True)


def test_scheduler_tables_and_steps_bit_exact():
    fx = load_fixture("schedulers")
    mo, xs = fx["model_output"], fx["sample"]
    for sname, e in fx["tables"].items():
        b, a, ac = R.noise_schedule(sname, 1000, **e["kw"])
        assert torch.equal(b, e["betas"]) and torch.equal(a, e["alphas"]) and torch.equal(ac, e["alphas_cumprod"])
        assert torch.equal(R.inference_timesteps(1000, 50), e["timesteps50"])
        for (pt, clip, t, eta), (prev, x0) in e["ddim"].items():
            noise = None
            if eta > 0:
                noise = torch.randn(mo.shape, dtype=mo.dtype, generator=torch.Generator().manual_seed(fx["noise_seed"]))
            p2, x2 = R.ddim_step(ac, 1000, 50, mo, t, xs, eta=eta, prediction_type=pt, clip_sample=clip, noise=noise)
            assert _eq(p2, prev) and _eq(x2, x0), (sname, pt, clip, t, eta)
        for (pt, vt, t), (prev, x0) in e["ddpm"].items():
            m = fx["model_output2"] if vt.startswith("learned") else mo
            shape = list(m.shape)
            if vt.startswith("learned"):
                shape[1] //= 2
            noise = torch.randn(shape, dtype=m.dtype, generator=torch.Generator().manual_seed(fx["noise_seed"]))
            p2, x2 = R.ddpm_step(b, a, ac, m, t, xs, prediction_type=pt, variance_type=vt, noise=noise)
            assert _eq(p2, prev) and _eq(x2, x0), (sname, pt, vt, t)
        ts = torch.tensor([999, 3])
        assert _eq(R.add_noise(ac, xs, mo, ts), e["add_noise"])
        assert _eq(R.get_velocity(ac, xs, mo, ts), e["get_velocity"])


def test_ddim_chain_matches_reference():
    fx = load_fixture("chain_c1a3d")
    _, _, ac = R.noise_schedule("scaled_linear_beta", 1000, beta_start=0.0005, beta_end=0.0195)
    sched = dict(alphas_cumprod=ac, num_train_timesteps=1000, num_inference_steps=10,
                 timesteps=R.inference_timesteps(1000, 10), clip_sample=False)
    inter = []
    with torch.no_grad():
        out = R.ddim_sample(fx["state_dict"], fx["cfg"], fx["noise"], sched,
                            on_step=lambda t, im: inter.append(im) if t % 100 == 0 else None)
    # free-running clip_sample=False chain in fp32: reference self-noise 1.9e-4 on sigma~49
    scale = fx["ddim_out"].abs().max().item()
    assert_close(out, fx["ddim_out"], atol=2e-5 * max(1.0, scale), what="ddim chain")
    assert len(inter) == len(fx["ddim_inter"])


def test_pndm_tables_and_sequences_match_reference():
    """PNDMScheduler (pndm.py): timestep tables (50 -> 59, 100 -> 109 evaluations) and a model-free step sequence through every
    branch of the Runge-Kutta / multi-step state machine, bit-exact against the reference's outputs."""
    fx = load_fixture("pndm_likelihood")
    for (n, skip), e in fx["tables"].items():
        prk, plms, ts = R.pndm_timesteps(1000, n, skip_prk_steps=skip)
        assert torch.equal(torch.from_numpy(ts), e["timesteps"]) and len(ts) == e["num_inference_steps"], (n, skip)
        assert torch.equal(torch.from_numpy(plms.copy()), e["plms"]) and len(prk) == len(e["prk"])
    assert fx["tables"][(100, False)]["num_inference_steps"] == 109  # the reference's structural pin (test_scheduler_pndm.py:58-62)
    for (sname, pt, skip, one), e in fx["sequences"].items():
        _, _, ac = R.noise_schedule(sname, 1000, **e["kw"])
        s = R.PNDM(ac, 1000, e["steps"], skip_prk_steps=skip, set_alpha_to_one=one, prediction_type=pt)
        x = torch.randn(e["shape"], generator=torch.Generator().manual_seed(e["x0_seed"]))
        assert len(s.timesteps) == len(e["samples"])
        for k, t in enumerate(s.timesteps):
            mo = torch.randn(e["shape"], generator=torch.Generator().manual_seed(e["mo_seed0"] + k))
            x = s.step(mo, int(t), x)
            assert _eq(x, e["samples"][k]), (sname, pt, skip, one, k)


def test_pndm_chain_and_likelihood_match_reference():
    fx = load_fixture("pndm_likelihood")
    c = fx["chain"]
    _, _, ac = R.noise_schedule("scaled_linear_beta", 1000, beta_start=0.0005, beta_end=0.0195)
    s = R.PNDM(ac, 1000, c["steps"])
    x = c["noise"]
    with torch.no_grad():
        for t in s.timesteps:
            x = s.step(R.unet_forward(c["state_dict"], c["cfg"], x, torch.Tensor((int(t),))), int(t), x)
    assert_close(x, c["out"], atol=2e-5 * max(1.0, c["out"].abs().max().item()), what="pndm chain")
    lk = fx["likelihood"]
    b, a, ac = R.noise_schedule("linear_beta", 10)
    torch.manual_seed(lk["noise_seed"])
    noise = torch.randn_like(lk["inputs"])
    model = lambda xx, ts, ctx: R.unet_forward(lk["state_dict"], lk["cfg"], xx, ts, ctx)  # noqa: E731
    for (vt, pt, clip), e in lk["cases"].items():
        with torch.no_grad():
            total, maps = R.get_likelihood(model, lk["inputs"], noise, b, a, ac, torch.arange(9, -1, -1), prediction_type=pt,
                                           variance_type=vt, clip_sample=clip)
        assert_close(total, e["total"], atol=1e-5 * e["total"].abs().max().item(), what=f"likelihood total {vt} {pt} {clip}")
        for got, want in zip(maps, e["maps"]):
            assert_close(got, want, atol=2e-5 * max(1.0, want.abs().max().item()), what=f"kl map {vt} {pt} {clip}")


def test_controlnet_forward_and_conditioned_chain_match_reference():
    """ControlNet (controlnet.py:367-436) and the ControlNet-conditioned DDIM chain (inferer.py:565-707) restated vs the reference."""
    fx = load_fixture("controlnet")
    for name, e in fx["forwards"].items():
        with torch.no_grad():
            down, mid = R.controlnet_forward(e["state_dict"], e["cfg"], e["x"], e["timesteps"], e["cond"], e["scale"], e["context"], e["class_labels"])
        assert len(down) == len(e["down"])
        for a, b in zip(down, e["down"]):
            assert_close(a, b, atol=2e-5, what=f"{name} down residual")
        assert_close(mid, e["mid"], atol=2e-5, what=f"{name} mid residual")
    i = fx["inferer"]
    _, _, ac = R.noise_schedule("scaled_linear_beta", 1000, beta_start=0.0005, beta_end=0.0195)
    img = i["noise"]
    with torch.no_grad():
        for t in R.inference_timesteps(1000, i["steps"]):
            ts = torch.Tensor((int(t),))
            down, mid = R.controlnet_forward(i["cn_sd"], i["cn_cfg"], img, ts, i["cond"])
            eps = R.unet_forward(i["unet_sd"], i["unet_cfg"], img, ts, None, None, down, mid)
            img, _ = R.ddim_step(ac, 1000, i["steps"], eps, int(t), img, clip_sample=False)
        assert_close(img, i["chain"], atol=2e-5 * max(1.0, i["chain"].abs().max().item()), what="controlnet chain")
        noisy = R.add_noise(ac, i["call_inputs"], i["noise"], i["call_timesteps"])
        down, mid = R.controlnet_forward(i["cn_sd"], i["cn_cfg"], noisy, i["call_timesteps"], i["cond"])
        pred = R.unet_forward(i["unet_sd"], i["unet_cfg"], noisy, i["call_timesteps"], None, None, down, mid)
    assert_close(pred, i["call_prediction"], atol=2e-5, what="controlnet inferer __call__")


def test_spade_networks_match_reference():
    """SPADE (spade_norm.py:79-96), SPADEDiffusionModelUNet (spade_diffusion_model_unet.py:836-912), SPADEAutoencoderKL
    (spade_autoencoderkl.py:410-484) and the seg-conditioned latent DDIM chain (inferer.py:364-487) restated vs the reference."""
    fx = load_fixture("spade")
    for name, e in fx["blocks"].items():
        kw = e["kwargs"]
        groups = kw.get("norm_params", {}).get("num_groups", kw["norm_nc"])  # instance norm = one group per channel
        eps = kw.get("norm_params", {}).get("eps", 1e-5)
        with torch.no_grad():
            y = R.spade({"." + k: v for k, v in e["state_dict"].items()}, "", e["x"], e["seg"], groups, eps)
        assert_close(y, e["y"], atol=2e-5 * max(1.0, e["y"].abs().max().item()), what=f"spade block {name}")
    for name, e in fx["unets"].items():
        with torch.no_grad():
            y = R.unet_forward(e["state_dict"], e["cfg"], e["x"], e["timesteps"], e["context"], seg=e["seg"])
        assert_close(y, e["y"], atol=2e-5, what=name)
    for name, e in fx["aekls"].items():
        with torch.no_grad():
            z_mu, z_sigma = R.aekl_encode(e["state_dict"], e["cfg"], e["x"])
            dec = R.aekl_decode(e["state_dict"], e["cfg"], e["z_mu"], e["seg"])
        assert_close(z_mu, e["z_mu"], atol=2e-5, what=f"{name} z_mu")
        assert_close(dec, e["decoded"], atol=2e-5 * max(1.0, e["decoded"].abs().max().item()), what=f"{name} decode")
    l = fx["latent"]
    _, _, ac = R.noise_schedule("scaled_linear_beta", 1000, beta_start=0.0005, beta_end=0.0195)
    img = l["noise"]
    with torch.no_grad():
        for t in R.inference_timesteps(1000, l["steps"]):
            eps_ = R.unet_forward(l["unet_sd"], l["unet_cfg"], img, torch.Tensor((int(t),)), seg=l["seg"])
            img, _ = R.ddim_step(ac, 1000, l["steps"], eps_, int(t), img, clip_sample=False)
        out = R.aekl_decode(l["ae_sd"], l["ae_cfg"], img / l["scale_factor"], l["seg"])
    assert_close(out, l["image"], atol=2e-5 * max(1.0, l["image"].abs().max().item()), what="spade latent chain")


def test_transformer_ordering_and_vqvae_transformer_inferer_match_reference():
    """DecoderOnlyTransformer (transformer.py:98-106), Ordering (ordering.py), VQVAETransformerInferer __call__ / get_likelihood /
    greedy sample (inferer.py:1126-1330) restated vs the reference's outputs."""
    fx = load_fixture("transformer")
    for name, e in fx["forwards"].items():
        with torch.no_grad():
            y = R.transformer_forward(e["state_dict"], e["cfg"], e["tokens"], e["context"])
        assert_close(y, e["logits"], atol=2e-5, what=f"transformer {name}")
    for key, e in fx["orderings"].items():
        order, revert = R.ordering_indices(**e["kw"])
        assert torch.equal(torch.as_tensor(order.copy()), e["order"]) and torch.equal(torch.as_tensor(revert.copy()), e["revert"]), key
    i = fx["inferer"]
    order, revert = R.ordering_indices(**i["ordering"])
    with torch.no_grad():
        idx = R.vq_index_quantize(i["vq_sd"], R.vqvae_encode(i["vq_sd"], i["vq_cfg"], i["x"]))[0]
        lat = idx.reshape(2, -1)[:, order]
        assert torch.equal(lat, i["target"])
        seq = torch.nn.functional.pad(lat, (1, 0), "constant", 16)[:, :-1].long()
        assert_close(R.transformer_forward(i["tr_sd"], i["tr_cfg"], seq), i["prediction"], atol=2e-5, what="inferer __call__")
        lik = R.transformer_likelihood(i["tr_sd"], i["tr_cfg"], idx, order, revert, 16)
        assert_close(lik, i["likelihood"], atol=2e-5, what="transformer likelihood")
        # greedy sampling (top_k = 1): the argmax path of the restated sampling head reproduces the reference's image
        s = torch.full((2, 1), 16).long()
        for _ in range(16):
            logits = R.transformer_forward(i["tr_sd"], i["tr_cfg"], s[:, -16:])[:, -1, :]
            s = torch.cat([s, R.transformer_sample_probs(logits, 1.0, 1, 16).argmax(-1, keepdim=True)], 1)
        img = R.vqvae_decode(i["vq_sd"], i["vq_cfg"], R.vq_embed(i["vq_sd"], s[:, 1:][:, revert].reshape(2, 4, 4)))
        assert_close(img, i["greedy_image"], atol=2e-5, what="greedy sample")
        w = fx["window"]
        idx2 = R.vq_index_quantize(i["vq_sd"], R.vqvae_encode(i["vq_sd"], i["vq_cfg"], w["x"]))[0]
        o2, r2 = R.ordering_indices("raster_scan", 2, (1, 2, 2))
        assert_close(R.transformer_likelihood(w["tr_sd"], w["tr_cfg"], idx2, o2, r2, 16), w["likelihood"], atol=2e-5, what="windowed likelihood")


def _grid_sd(e):
    return R.synthetic_state_dict(e["shapes"], seed=e["seed"])


def test_constructor_argument_grid_matches_reference():
    """The restatement over a grid of constructor arguments in the spirit of the reference's shape tests (res-block tuples,
    resblock_updown, attention placements / head widths, cross-attention depth, class embeddings, non-local attention switches,
    ConvTranspose up-sampling, scalar-vs-tuple VQ-VAE parameters; 2-D and 3-D) against reference outputs (oracle/make_golden.py
    --grid-only)."""
    fx = load_fixture("config_grid")
    with torch.no_grad():
        for e in fx["unet"]:
            y = R.unet_forward(_grid_sd(e), e["cfg"], e["x"], e["timesteps"], e["context"], e["class_labels"])
            assert_close(y, e["y"], atol=2e-5 * max(1.0, e["y"].abs().max().item()), what=f"unet {e['cfg']}")
        for e in fx["aekl"]:
            sd = _grid_sd(e)
            mu, sigma = R.aekl_encode(sd, e["cfg"], e["x"])
            assert_close(mu, e["z_mu"], atol=2e-5 * max(1.0, e["z_mu"].abs().max().item()), what=f"aekl mu {e['cfg']}")
            assert_close(sigma, e["z_sigma"], atol=2e-5 * max(1.0, e["z_sigma"].abs().max().item()), what=f"aekl sigma {e['cfg']}")
            assert_close(R.aekl_decode(sd, e["cfg"], e["z_mu"]), e["reconstruction"], atol=2e-5 * max(1.0, e["reconstruction"].abs().max().item()),
                         what=f"aekl decode {e['cfg']}")
        for e in fx["vqvae"]:
            sd = _grid_sd(e)
            z = R.vqvae_encode(sd, e["cfg"], e["x"])
            assert_close(z, e["z"], atol=2e-5 * max(1.0, e["z"].abs().max().item()), what=f"vqvae z {e['cfg']}")
            assert torch.equal(R.vq_index_quantize(sd, e["z"])[0], e["indices"])
            assert_close(R.vqvae_decode(sd, e["cfg"], R.vq_embed(sd, e["indices"])), e["reconstruction"],
                         atol=2e-5 * max(1.0, e["reconstruction"].abs().max().item()), what=f"vqvae decode {e['cfg']}")


def test_restatement_matches_the_reference_at_the_headline_size():
    """The oracle pinned at BASELINE's REAL size: oracle/restatement.py's C2 forward of the benchmark's own 1x1x128^3 noise volume at t = 500 against the
    unmodified reference's output (tests/golden/c2_fullsize_ref.pt: every-4th-voxel lattice + whole-tensor summaries; about a minute on 8 cores)."""
    import os

    from _util import GOLDEN
    from bench import C2, rerandomize_zero_params
    from generativemodels_amd.networks.nets import DiffusionModelUNet
    from make_golden_c2_fullsize import sd_checksum, whole_tensor_summary

    fx = torch.load(os.path.join(GOLDEN, "c2_fullsize_ref.pt"), weights_only=False)
    torch.manual_seed(0)
    sd = rerandomize_zero_params({k: v.clone() for k, v in DiffusionModelUNet(**C2).eval().state_dict().items()})
    assert abs(sd_checksum(sd) - fx["sd_checksum"]) <= 1e-9 * fx["sd_checksum"]
    x = torch.randn((1, 1, 128, 128, 128), generator=torch.Generator().manual_seed(fx["input_seed"]))
    with torch.no_grad():
        y = R.unet_forward(sd, C2, x, torch.tensor([500.0]))
    ref, L = fx["fp32"][500], fx["lattice"]
    err = (y[..., ::L, ::L, ::L] - ref["lattice"]).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref["absmax"]), err  # same algorithm, same fp32 kernels of the same torch: only thread-partition round-off differs
    got = whole_tensor_summary(y)
    assert abs(got["std"] - ref["std"]) <= 1e-5 * ref["std"] and abs(got["absmax"] - ref["absmax"]) <= 1e-4 * ref["absmax"]
