"""GPU (-m gpu): the drop-in modules (DiffusionModelUNet, AutoencoderKL, VQVAE, schedulers, inferers) against

  (1) golden outputs of the UNMODIFIED reference (tests/golden/*.pt, produced by oracle/make_golden.py), and
  (2) the CPU oracle (oracle/restatement.py) on fresh seeded inputs, incl. a free-running DDIM chain and a seeded DDPM chain.

fp32 tolerance: max|err| <= 1e-4 * max(1, |ref|_inf) (the reference's own fp32-vs-fp64 noise is 1.2e-5).
bf16 tolerance (vs the fp32 reference): mean|err| <= 2e-2 * sigma and max|err| <= 0.2 * sigma."""
import pytest
import torch

import restatement as R
from _util import cast_sd, load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _nets():
    from generativemodels_amd.networks import nets
    return nets


def _ops():
    from generativemodels_amd import ops
    return ops


def _fp32_close(got, want, what, factor=1.0):
    got, want = got.double().cpu(), want.double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    tol = 1e-4 * max(1.0, want.abs().max().item()) * factor
    err = (got - want).abs().max().item()
    assert err <= tol, f"{what}: max|err| {err:.3e} > {tol:.3e}"


def _bf16_close(got, want, what):
    got, want = got.double().cpu(), want.double().cpu()
    sigma = max(want.std().item(), 1e-3)
    err = (got - want).abs()
    assert err.mean().item() <= 2e-2 * sigma and err.max().item() <= 0.2 * sigma, \
        f"{what}: mean|err| {err.mean().item():.3e}, max|err| {err.max().item():.3e}, sigma {sigma:.3e}"


def _dev(x):
    return None if x is None else x.to(DEV)


UNETS = ["unet2d_c1a", "unet3d_c1a", "unet2d_c1b", "unet3d_c2mini", "unet2d_cond", "unet3d_cond"]


def _build_unet(fx, dtype=torch.float32):
    m = _nets().DiffusionModelUNet(**fx["cfg"]).eval()
    m.load_state_dict(fx["state_dict"], strict=True)
    return m.to(DEV, dtype)


@pytest.mark.parametrize("name", UNETS)
def test_unet_fp32_matches_reference_golden(name):
    fx = load_fixture(name)
    i = fx["inputs"]
    m = _build_unet(fx)
    y = m(_dev(i["x"]), _dev(i["timesteps"]), context=_dev(i["context"]), class_labels=_dev(i["class_labels"]))
    _fp32_close(y, fx["outputs"]["y"], name)
    # (1,)-shaped shared timestep broadcasts over the batch exactly like `sample` does (inferer.py:129,133)
    if i["x"].shape[0] > 1 and i["class_labels"] is None:
        t1 = i["timesteps"][:1]
        with torch.no_grad():
            want = R.unet_forward(fx["state_dict"], fx["cfg"], i["x"], t1, i["context"], None)
        _fp32_close(m(_dev(i["x"]), _dev(t1), context=_dev(i["context"])), want, name + " shared timestep")


@pytest.mark.parametrize("name", UNETS)
def test_unet_bf16_close_to_fp32_reference(name):
    """bf16 bar of SURVEY.md 8(c)(3): absolute (vs the fp32 golden) AND relative to the reference's own bf16 run
    (tests/golden/unet_bf16_ref.pt, oracle/make_golden.py --bf16-only): mean|err| <= 1.5 x the reference's bf16 mean|err|."""
    fx = load_fixture(name)
    i = fx["inputs"]
    m = _build_unet(fx, torch.bfloat16)
    ctx = None if i["context"] is None else i["context"].bfloat16()
    y = m(_dev(i["x"].bfloat16()), _dev(i["timesteps"]), context=_dev(ctx), class_labels=_dev(i["class_labels"]))
    assert y.dtype == torch.bfloat16
    _bf16_close(y, fx["outputs"]["y"], name)
    ref = load_fixture("unet_bf16_ref")["cases"][name]
    err = (y.float().cpu() - fx["outputs"]["y"]).abs()
    assert err.mean().item() <= 1.5 * ref["mean_err"], f"{name}: ours bf16 mean|err| {err.mean().item():.3e} vs reference bf16 {ref['mean_err']:.3e}"
    assert err.max().item() <= 2.0 * ref["max_err"], f"{name}: ours bf16 max|err| {err.max().item():.3e} vs reference bf16 {ref['max_err']:.3e}"


def test_unet_forward_errors_match_reference():
    fx = load_fixture("unet2d_c1a")
    m = _build_unet(fx)
    x = _dev(fx["inputs"]["x"])
    with pytest.raises(ValueError):
        m(x, torch.zeros((2, 1), device=DEV))  # timesteps must be 1-D (diffusion_model_unet.py:471-472)
    with pytest.raises(ValueError):
        m(x, _dev(fx["inputs"]["timesteps"]), context=torch.zeros((2, 1, 3), device=DEV))  # context without conditioning
    with pytest.raises(RuntimeError):
        m(fx["inputs"]["x"], fx["inputs"]["timesteps"])  # CPU tensors: no fallback
    fxc = load_fixture("unet2d_cond")
    mc = _build_unet(fxc)
    with pytest.raises(ValueError):
        mc(_dev(fxc["inputs"]["x"]), _dev(fxc["inputs"]["timesteps"]), context=_dev(fxc["inputs"]["context"]))  # class_labels


@pytest.mark.parametrize("name", ["aekl2d", "aekl3d_brainlike", "aekl3d_convT"])
def test_autoencoderkl_matches_reference_golden(name):
    fx = load_fixture(name)
    m = _nets().AutoencoderKL(**fx["cfg"]).eval()
    m.load_state_dict(fx["state_dict"], strict=True)
    m = m.to(DEV)
    x, o = fx["inputs"]["x"], fx["outputs"]
    mu, sigma = m.encode(_dev(x))
    _fp32_close(mu, o["z_mu"], "z_mu")
    _fp32_close(sigma, o["z_sigma"], "z_sigma")
    _fp32_close(m.decode(_dev(o["z_mu"])), o["reconstruction"], "reconstruction")
    _fp32_close(m.reconstruct(_dev(x)), o["reconstruction"], "reconstruct", factor=2.0)
    rec, mu2, sg2 = m(_dev(x))
    assert rec.shape == o["reconstruction"].shape and torch.equal(mu2, mu) and torch.equal(sg2, sigma)
    z = m.encode_stage_2_inputs(_dev(x))
    assert z.shape == mu.shape and torch.isfinite(z).all()
    # sampling() = mu + eps * sigma with eps ~ N(0, 1): with sigma = 0 it must return mu exactly
    assert torch.equal(m.sampling(mu, torch.zeros_like(sigma)), mu)
    mb = _nets().AutoencoderKL(**fx["cfg"]).eval()
    mb.load_state_dict(fx["state_dict"])
    mb = mb.to(DEV, torch.bfloat16)
    _bf16_close(mb.decode(_dev(o["z_mu"].bfloat16())), o["reconstruction"], "bf16 reconstruction")


@pytest.mark.parametrize("name", ["vqvae3d", "vqvae2d_odd"])
def test_vqvae_matches_reference_golden(name):
    fx = load_fixture(name)
    m = _nets().VQVAE(**fx["cfg"]).eval()
    m.load_state_dict(fx["state_dict"], strict=True)
    m = m.to(DEV)
    x, o = fx["inputs"]["x"], fx["outputs"]
    _fp32_close(m.encode(_dev(x)), o["z"], "z")
    q, loss = m.quantize(_dev(o["z"]))
    _fp32_close(q, o["quantized"], "quantized")
    assert abs(loss.item() - o["loss"].item()) <= 1e-6
    assert torch.equal(m.quantizer.quantize(_dev(o["z"])).cpu(), o["indices"])
    assert torch.equal(m.index_quantize(_dev(x)).cpu(), o["indices"])  # integer work: bit-exact
    _fp32_close(m.decode(_dev(o["quantized"])), o["reconstruction"], "reconstruction")
    _fp32_close(m.decode_samples(_dev(o["indices"])), o["reconstruction"], "decode_samples")
    rec, l2 = m(_dev(x))
    _fp32_close(rec, o["reconstruction"], "forward")
    assert abs(l2.item() - o["loss"].item()) <= 1e-6
    assert float(m.quantizer.perplexity) >= 1.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ema_quantizer_training_forward_matches_reference(dtype):
    """EMAQuantizer in train() mode: code lookup, commitment loss, the EMA codebook update (gm_vq_ema_stats -> one flat buffer ->
    gm_vq_ema_update) carried over two steps, and the straight-through / loss gradient in the input -- against the outputs of the
    unmodified reference (tests/golden/vq_ema.pt; vector_quantizer.py:161-188).  bf16: the module keeps bf16 tables like the reference
    would; checked against the fp32 golden with a bf16-sized tolerance."""
    from generativemodels_amd.networks.layers.vector_quantizer import EMAQuantizer
    fx = load_fixture("vq_ema")
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    for name, case in fx["cases"].items():
        layer = EMAQuantizer(**case["args"])
        layer.load_state_dict(case["init"])
        layer = layer.to(DEV, dtype).train()
        for s, st in enumerate(case["steps"]):
            x = st["x"].to(DEV, dtype).requires_grad_(True)
            q, loss, idx = layer(x)
            ((q.float() * st["gq"].to(DEV)).sum() + 3.0 * loss.float()).backward()
            if dtype == torch.float32:
                assert torch.equal(idx.cpu(), st["indices"]), (name, s)
            scale = max(1.0, st["quantized"].abs().max().item())
            if dtype == torch.float32:
                assert (q.float().cpu() - st["quantized"]).abs().max().item() <= tol * scale
                assert abs(loss.item() - st["loss"].item()) <= tol
                assert (x.grad.float().cpu() - st["dx"]).abs().max().item() <= tol * max(1.0, st["dx"].abs().max().item())
                for k, v in layer.state_dict().items():
                    assert (v.float().cpu() - st["state"][k]).abs().max().item() <= tol * max(1.0, st["state"][k].abs().max().item()), (name, s, k)
            else:  # bf16 inputs / tables: near-ties may pick another code; the statistics stay close
                assert (idx.cpu() == st["indices"]).float().mean().item() > 0.9
                assert abs(loss.item() - st["loss"].item()) <= 0.1 * max(st["loss"].item(), 1e-3)
                assert torch.isfinite(x.grad.float()).all() and torch.isfinite(layer.embedding.weight.float()).all()
        # eval() afterwards: the plain inference path on the updated codebook, no state change
        layer.eval()
        before = {k: v.clone() for k, v in layer.state_dict().items()}
        layer(case["steps"][0]["x"].to(DEV, dtype))
        assert all(torch.equal(before[k], v) for k, v in layer.state_dict().items())


def test_ddim_chain_free_running_matches_reference():
    """10-step free-running DDIM chain (clip_sample=False) of the literal reference test model; the reference's own fp32
    self-noise on this chain is ~2e-5 * scale (tests/test_oracle_golden.py)."""
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.schedulers import DDIMScheduler
    fx = load_fixture("chain_c1a3d")
    m = _build_unet(fx)
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(10)
    for graph in (False, True):
        inf = DiffusionInferer(sched, use_hip_graph=graph)
        out, inter = inf.sample(_dev(fx["noise"]), m, sched, save_intermediates=True, intermediate_steps=100, verbose=False)
        _fp32_close(out, fx["ddim_out"], f"ddim chain (graph={graph})", factor=2.0)
        assert len(inter) == len(fx["ddim_inter"])
        for a, b in zip(inter, fx["ddim_inter"]):
            _fp32_close(a, b, "ddim intermediate", factor=2.0)


def test_captured_forward_is_reused_across_sample_calls_and_refreshed_when_parameters_change():
    """`use_hip_graph=True`: the captured UNet forward is kept on the inferer and serves later sample() calls (a capture costs two eager
    forwards -- 7 % of a C3 sample); a parameter update between calls (its `_version` moves) re-captures, because the packed MFMA panels a
    capture baked in are derived per version.  Every graphed chain equals the eager chain bit for bit."""
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.schedulers import DDIMScheduler
    fx = load_fixture("chain_c1a3d")
    m = _build_unet(fx)
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(4)
    eager, graphed = DiffusionInferer(sched, use_hip_graph=False), DiffusionInferer(sched, use_hip_graph=True)
    noise = _dev(fx["noise"])
    want = eager.sample(noise, m, sched, verbose=False)
    got1 = graphed.sample(noise, m, sched, verbose=False)
    g1 = graphed._graph_cache[-1][3]
    got2 = graphed.sample(noise * 0.5, m, sched, verbose=False)
    assert graphed._graph_cache[-1][3] is g1 and len(graphed._graph_cache) == 1   # second call: the same capture
    assert torch.equal(got1, want) and torch.equal(got2, eager.sample(noise * 0.5, m, sched, verbose=False))
    with torch.no_grad():  # an "optimizer step": in-place update bumps the version
        for p in m.parameters():
            p.mul_(1.01)
    got3 = graphed.sample(noise, m, sched, verbose=False)
    assert graphed._graph_cache[-1][3] is not g1 and len(graphed._graph_cache) == 1  # re-captured, the stale capture dropped
    want3 = eager.sample(noise, m, sched, verbose=False)
    assert torch.equal(got3, want3) and not torch.equal(got3, want)
    # an EMA-weight swap through .data: no _version moves, the storage does -- the stale capture (old panels, old bias storage) must not be replayed
    g3 = graphed._graph_cache[-1][3]
    for p in m.parameters():
        p.data = (p.data * 0.97).clone()
    got4 = graphed.sample(noise, m, sched, verbose=False)
    assert graphed._graph_cache[-1][3] is not g3
    want4 = eager.sample(noise, m, sched, verbose=False)
    assert torch.equal(got4, want4) and not torch.equal(got4, want3)


def test_bf16_ddpm_chain_is_the_same_eager_graphed_and_through_torchs_own_noise_draw(monkeypatch):
    """(round 6) A seeded bf16 DDPM chain of BASELINE configs[0]'s UNet (2-D, attention at level 1): eager launches, the HIP-graph replay of the forward, and the
    chain whose noise goes through torch's own bf16 `randn` (host_noise.ENABLED = False: no byte draws, no table, no pinned staging, no lookup inside the step's
    kernel) must produce the SAME images bit for bit and leave the CPU generator in the same state (reference: inferer.py:83-143 + ddpm.py:191-252)."""
    from generativemodels_amd import host_noise as H
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.schedulers import DDPMScheduler
    cfg = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1, num_head_channels=64)
    torch.manual_seed(0)
    m = _nets().DiffusionModelUNet(**cfg).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    R.derandomize_zeros(sd)
    m.load_state_dict(sd)
    m = m.to(DEV, torch.bfloat16)
    sched = DDPMScheduler(1000)
    sched.set_timesteps(8)
    noise = torch.randn((4, 1, 64, 64), generator=torch.Generator().manual_seed(7)).to(DEV, torch.bfloat16)

    def chain(graph):
        torch.manual_seed(3)
        out = DiffusionInferer(sched, use_hip_graph=graph).sample(noise, m, sched, verbose=False)
        return out, torch.get_rng_state()

    (a, sa), (b, sb) = chain(False), chain(True)
    monkeypatch.setattr(H, "ENABLED", False)
    (c, sc), (d, sd_) = chain(False), chain(True)
    assert torch.isfinite(a.float()).all()
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    assert torch.equal(sa, sb) and torch.equal(sa, sc) and torch.equal(sa, sd_)


def test_sampling_loop_takes_its_timestep_rows_from_one_batched_pass():
    """(round 5) `DiffusionInferer.sample` computes the timestep rows of the whole chain once (`DiffusionModelUNet.time_rows_table`: embedding, two-layer
    MLP and the stacked `time_emb_proj` GEMM over all T timesteps -- 4 launches per chain instead of 4 per step; reference:
    diffusion_model_unet.py:1895-1905, :686-690 once per step) and hands row i to step i.  The table equals the per-step rows to fp32 rounding
    (another GEMM row count = another summation order), a chain through sample() equals the manual chain of single forwards to the same bar, nothing
    is left on the model afterwards, and a class-conditional model (rows depend on the labels) takes the per-step path."""
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.nets import DiffusionModelUNet
    from generativemodels_amd.networks.schedulers import DDIMScheduler
    fx = load_fixture("chain_c1a3d")
    m = _build_unet(fx)
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(5)
    steps = [int(t) for t in sched.timesteps.tolist()]
    t_dev = torch.tensor(steps, dtype=torch.float32, device=DEV)
    table = m.time_rows_table(t_dev)
    assert table is not None and table.shape[0] == len(steps) and table.dtype == torch.float32
    for i in range(len(steps)):
        one = m._temb_stacked(t_dev[i:i + 1], None)
        scale = max(1.0, one.abs().max().item())
        assert (table[i:i + 1] - one).abs().max().item() <= 2e-6 * scale, i
    noise = _dev(fx["noise"])
    inf = DiffusionInferer(sched, use_hip_graph=False)
    got = inf.sample(noise, m, sched, verbose=False)
    assert "_time_rows_row" not in m.__dict__
    x = noise
    with torch.no_grad():
        for i, t in enumerate(steps):
            x, _ = sched.step(m(x, timesteps=t_dev[i:i + 1]), t, x)
    _fp32_close(got, x.cpu(), "sample() with the batched timestep rows vs single forwards", factor=1.0)
    # a row handed over and never consumed must not leak into a later direct call
    m._time_rows_row = table[0:1]
    try:
        inf.sample(noise[:, :, :1], m, sched, verbose=False)  # wrong shape: raises inside the first forward
    except Exception:
        pass
    assert "_time_rows_row" not in m.__dict__
    torch.manual_seed(3)
    mc = DiffusionModelUNet(spatial_dims=2, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(32, 32), attention_levels=(False, False),
                            norm_num_groups=32, num_class_embeds=4).to(DEV).eval()
    assert mc.time_rows_table(t_dev) is None


def test_ddpm_chain_with_seeded_cpu_noise_matches_reference():
    """DDPM draws its noise from the global CPU generator (ddpm.py:244-247): same seed => same chain as the reference."""
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.schedulers import DDPMScheduler
    fx = load_fixture("chain_c1a3d")
    m = _build_unet(fx)
    sched = DDPMScheduler(num_train_timesteps=10)
    sched.set_timesteps(10)
    torch.manual_seed(fx["ddpm_global_seed"])
    out, inter = DiffusionInferer(sched).sample(_dev(fx["noise"]), m, sched, save_intermediates=True, intermediate_steps=1, verbose=False)
    assert len(inter) == len(fx["ddpm_inter"]) == 10
    _fp32_close(out, fx["ddpm_out"], "ddpm chain", factor=5.0)  # clip_sample=True chain: clamps amplify rounding at the knees
    pred = DiffusionInferer(sched)(inputs=_dev(fx["call_inputs"]), diffusion_model=m, noise=_dev(fx["noise"]),
                                   timesteps=_dev(fx["call_timesteps"]))
    _fp32_close(pred, fx["call_prediction"], "inferer __call__")


def test_ddpm_fp32_noise_draw_for_reduced_precision_chains():
    """`DDPMScheduler.fp32_noise_draw` (round 6; not in the reference): a bf16 chain draws its noise as fp32 values from the CPU generator and rounds them on the
    device, instead of torch's serial bf16 fill (0.9 ms per 16 x 1 x 64 x 64 draw on the GPU box's host: more than the replayed bf16 forward).  With the switch the
    bf16 step sees the SAME noise as the fp32 step of the same seed (rounded); without it the bf16 draw is another stream.  fp32 steps do not change."""
    from generativemodels_amd.networks.schedulers import DDPMScheduler
    sched = DDPMScheduler(num_train_timesteps=100)
    sched.set_timesteps(100)
    eps = torch.randn((4, 1, 16, 16), generator=torch.Generator().manual_seed(1))
    x = torch.randn((4, 1, 16, 16), generator=torch.Generator().manual_seed(2))
    torch.manual_seed(9)
    ref32, _ = sched.step(_dev(eps), 50, _dev(x))
    try:
        DDPMScheduler.fp32_noise_draw = True
        torch.manual_seed(9)
        again32, _ = sched.step(_dev(eps), 50, _dev(x))
        assert torch.equal(again32, ref32)
        torch.manual_seed(9)
        fast16, _ = sched.step(_dev(eps.bfloat16()), 50, _dev(x.bfloat16()))
        state_fast = torch.get_rng_state()
    finally:
        DDPMScheduler.fp32_noise_draw = False
    torch.manual_seed(9)
    slow16, _ = sched.step(_dev(eps.bfloat16()), 50, _dev(x.bfloat16()))
    assert fast16.dtype == torch.bfloat16 and torch.isfinite(fast16.float()).all()
    assert (fast16.float() - ref32).abs().max().item() <= 3e-2           # the same noise, rounded: a bf16 step of the fp32 one
    assert (slow16.float() - ref32).abs().max().item() >= 1e-1           # torch's bf16 fill is another stream over the same generator
    torch.manual_seed(9)
    torch.randn((4, 1, 16, 16))
    assert torch.equal(state_fast, torch.get_rng_state())                # ... and the generator ends where one fp32 draw leaves it


def test_bf16_noise_of_an_ancestral_step_is_torch_randn_bit_for_bit(monkeypatch):
    """(round 6) A bf16 DDPM / DDIM(eta > 0) step draws its noise through host_noise.randn: the CPU generator's BYTE draws, copied to the device and expanded by
    gm_normal_bf16_from_bits -- against `torch.randn(shape, dtype=bfloat16, generator=g).to(device)`, the reference's draw (ddpm.py:244-248, ddim.py:231-234): the
    same bits, the generator left in the same state, for whole and ragged sizes (the latter take the plain draw); the steps with the table switched off are
    bit-identical."""
    from generativemodels_amd import host_noise as H
    from generativemodels_amd.networks.schedulers import DDIMScheduler, DDPMScheduler
    assert H.ENABLED and H.table_matches_torch()
    for seed, shape in ((3, (16, 1, 64, 64)), (4, (2, 4, 8, 8, 8)), (5, (3, 5, 7)), (6, (1, 16))):
        g1, g2 = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed)
        want = torch.randn(shape, dtype=torch.bfloat16, generator=g1).to(DEV)
        got = H.randn(shape, torch.bfloat16, g2, DEV)
        assert got.shape == want.shape and torch.equal(got.view(torch.int16), want.view(torch.int16)) and torch.equal(g1.get_state(), g2.get_state()), shape
    g1, g2 = torch.Generator().manual_seed(21), torch.Generator().manual_seed(21)
    for i in range(2 * H.RING + 1):  # the pinned staging ring wraps: every draw still is torch.randn's, fp32 and bf16 alike
        for dt in (torch.float32, torch.bfloat16):
            want = torch.randn((2, 1, 32, 32), dtype=dt, generator=g1).to(DEV)
            got = H.randn((2, 1, 32, 32), dt, g2, DEV)
            assert torch.equal(got, want), (i, dt)
    assert torch.equal(g1.get_state(), g2.get_state())
    # the device kernel against the host-side lookup on every byte pair
    pairs = torch.arange(65536, dtype=torch.int64)
    bits = torch.stack([pairs // 256, pairs % 256], 1).reshape(-1, 8, 2).permute(0, 2, 1).reshape(-1).to(torch.uint8)  # blocks of 16: 8 first bytes, 8 second bytes
    dev_out = _ops().normal_bf16_from_bits(bits.to(DEV), H.bf16_normal_table().to(DEV))
    assert torch.equal(dev_out.cpu().view(torch.int16), H.normal_bf16_from_bits_host(bits).view(torch.int16))
    eps = torch.randn((4, 1, 16, 16), generator=torch.Generator().manual_seed(1)).bfloat16()
    x = torch.randn((4, 1, 16, 16), generator=torch.Generator().manual_seed(2)).bfloat16()
    ddpm, ddim = DDPMScheduler(num_train_timesteps=100), DDIMScheduler(num_train_timesteps=100)
    ddpm.set_timesteps(100)
    ddim.set_timesteps(50)

    def steps():
        torch.manual_seed(9)
        a, _ = ddpm.step(_dev(eps), 50, _dev(x))
        b, _ = ddpm.step(_dev(eps), 49, a, generator=torch.Generator().manual_seed(11))
        c, _ = ddim.step(_dev(eps), 50, _dev(x), eta=0.7, generator=torch.Generator().manual_seed(12))
        return a, b, c, torch.get_rng_state()

    fast = steps()
    monkeypatch.setattr(H, "ENABLED", False)
    plain = steps()
    for f, p in zip(fast, plain):
        assert torch.equal(f, p)


def test_inferer_concat_mode_and_errors():
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.schedulers import DDIMScheduler
    cfg = dict(spatial_dims=2, in_channels=2, out_channels=1, num_channels=[8], norm_num_groups=8, attention_levels=[True],
               num_res_blocks=1, num_head_channels=8)
    torch.manual_seed(0)
    m = _nets().DiffusionModelUNet(**cfg).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    R.derandomize_zeros(sd)
    m.load_state_dict(sd)
    m = m.to(DEV)
    sched = DDIMScheduler(1000, clip_sample=False)
    sched.set_timesteps(4)
    noise = torch.randn((2, 1, 8, 8), generator=torch.Generator().manual_seed(3))
    cond = torch.randn((2, 1, 8, 8), generator=torch.Generator().manual_seed(4))
    inf = DiffusionInferer(sched)
    out = inf.sample(_dev(noise), m, sched, conditioning=_dev(cond), mode="concat", verbose=False)
    want = R.ddim_sample(sd, cfg, noise, dict(alphas_cumprod=sched.alphas_cumprod, num_train_timesteps=1000, num_inference_steps=4,
                                              timesteps=sched.timesteps, clip_sample=False), conditioning=cond, mode="concat")
    _fp32_close(out, want, "concat-mode chain", factor=2.0)
    with pytest.raises(NotImplementedError):
        inf.sample(_dev(noise), m, sched, mode="film", verbose=False)
    with pytest.raises(NotImplementedError):
        inf(inputs=_dev(noise), diffusion_model=m, noise=_dev(noise), timesteps=torch.tensor([1, 2]), mode="film")


def test_latent_diffusion_inferer_sample_and_call():
    from generativemodels_amd.inferers import LatentDiffusionInferer
    from generativemodels_amd.networks.schedulers import DDIMScheduler
    fxa = load_fixture("aekl3d_brainlike")
    ae = _nets().AutoencoderKL(**fxa["cfg"]).eval()
    ae.load_state_dict(fxa["state_dict"])
    ae = ae.to(DEV)
    ucfg = dict(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1,
                norm_num_groups=8, num_head_channels=(0, 16))
    torch.manual_seed(1)
    un = _nets().DiffusionModelUNet(**ucfg).eval()
    usd = {k: v.clone() for k, v in un.state_dict().items()}
    R.derandomize_zeros(usd)
    un.load_state_dict(usd)
    un = un.to(DEV)
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205, clip_sample=False)
    sched.set_timesteps(3)
    noise = torch.randn((1, 4, 4, 4, 4), generator=torch.Generator().manual_seed(9))
    inf = LatentDiffusionInferer(sched, scale_factor=0.8)
    img = inf.sample(_dev(noise), ae, un, sched, verbose=False)
    lat = R.ddim_sample(usd, ucfg, noise, dict(alphas_cumprod=sched.alphas_cumprod, num_train_timesteps=1000, num_inference_steps=3,
                                               timesteps=sched.timesteps, clip_sample=False))
    with torch.no_grad():
        want = R.aekl_decode(fxa["state_dict"], fxa["cfg"], lat / 0.8)
    _fp32_close(img, want, "latent sample", factor=2.0)
    x = fxa["inputs"]["x"]
    pred = inf(inputs=_dev(x), autoencoder_model=ae, diffusion_model=un, noise=_dev(torch.randn_like(noise)),
               timesteps=torch.tensor([5], device=DEV))
    assert tuple(pred.shape) == (1, 4, 2, 2, 2) or pred.shape[1] == 4
    with pytest.raises(ValueError):
        LatentDiffusionInferer(sched, ldm_latent_shape=[4, 4, 4])


def test_pndm_step_sequences_bit_exact_and_chain_matches_reference():
    """PNDMScheduler on the device: a model-free sequence through every branch of the RK / multi-step state machine is BIT-EXACT
    against the oracle run on this host (and within a few ulp-of-operand of the committed reference outputs, see the DDIM test for
    why golden vectors from another CPU are not bit-comparable); the UNet chain matches the reference's PNDM sample."""
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.schedulers import PNDMScheduler
    fx = load_fixture("pndm_likelihood")
    for (sname, pt, skip, one), e in fx["sequences"].items():
        s = PNDMScheduler(1000, schedule=sname, skip_prk_steps=skip, set_alpha_to_one=one, prediction_type=pt, **e["kw"])
        s.set_timesteps(e["steps"])
        o = R.PNDM(s.alphas_cumprod, 1000, e["steps"], skip_prk_steps=skip, set_alpha_to_one=one, prediction_type=pt)
        x0 = torch.randn(e["shape"], generator=torch.Generator().manual_seed(e["x0_seed"]))
        x, xo = _dev(x0), x0
        assert torch.equal(s.timesteps, o.timesteps)
        for k, t in enumerate(s.timesteps):
            mo = torch.randn(e["shape"], generator=torch.Generator().manual_seed(e["mo_seed0"] + k))
            x, none = s.step(_dev(mo), int(t), x)
            xo = o.step(mo, int(t), xo)
            assert none is None
            assert torch.equal(x.cpu(), xo), (sname, pt, skip, one, k, (x.cpu() - xo).abs().max().item())
            gold = e["samples"][k]
            assert (x.cpu() - gold).abs().max().item() <= 1e-5 * max(1.0, gold.abs().max().item()), (sname, pt, skip, one, k)
    c = fx["chain"]
    m = _nets().DiffusionModelUNet(**c["cfg"]).eval()
    m.load_state_dict(c["state_dict"])
    m = m.to(DEV)
    for graph in (False, True):
        s = PNDMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195)
        s.set_timesteps(c["steps"])
        out = DiffusionInferer(s, use_hip_graph=graph).sample(_dev(c["noise"]), m, s, verbose=False)
        _fp32_close(out, c["out"], f"pndm chain (graph={graph})", factor=2.0)
    # bf16 tensors: fp32 arithmetic, one rounding per fused op
    s = PNDMScheduler(1000)
    s.set_timesteps(20)
    o = R.PNDM(s.alphas_cumprod, 1000, 20)
    x0 = torch.randn((2, 2, 4, 4, 4), generator=torch.Generator().manual_seed(5))
    x, xo = _dev(x0.bfloat16()), x0.bfloat16().float()
    for k, t in enumerate(s.timesteps[:16]):
        mo = torch.randn((2, 2, 4, 4, 4), generator=torch.Generator().manual_seed(200 + k)).bfloat16()
        x, _ = s.step(_dev(mo), int(t), x)
        xo = o.step(mo.float(), int(t), xo)
        assert x.dtype == torch.bfloat16
    assert (x.float().cpu() - xo).abs().max().item() <= 0.05 * max(1.0, xo.abs().max().item())


def test_get_likelihood_matches_reference():
    """DiffusionInferer.get_likelihood (inferer.py:145-321): totals and per-step KL / decoder-NLL maps against the reference's
    outputs for every fixed-variance x prediction-type x clip combination, with the reference's seeded noise draw."""
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.schedulers import DDIMScheduler, DDPMScheduler
    lk = load_fixture("pndm_likelihood")["likelihood"]
    m = _nets().DiffusionModelUNet(**lk["cfg"]).eval()
    m.load_state_dict(lk["state_dict"])
    m = m.to(DEV)
    torch.manual_seed(lk["noise_seed"])
    noise = torch.randn_like(lk["inputs"])
    for (vt, pt, clip), e in lk["cases"].items():
        d = DDPMScheduler(num_train_timesteps=10, variance_type=vt, prediction_type=pt, clip_sample=clip)
        total, maps = DiffusionInferer(d).get_likelihood(_dev(lk["inputs"]), m, d, save_intermediates=True, verbose=False,
                                                         _noise=_dev(noise))
        assert total.shape == (2,) and total.dtype == torch.float32 and len(maps) == 10 and not maps[0].is_cuda
        tol = 1e-4 * e["total"].abs().max().item()
        assert (total.cpu() - e["total"]).abs().max().item() <= tol, (vt, pt, clip, total.cpu(), e["total"])
        for k, (got, want) in enumerate(zip(maps, e["maps"])):
            assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), (vt, pt, clip, k)
        only_total = DiffusionInferer(d).get_likelihood(_dev(lk["inputs"]), m, d, verbose=False, _noise=_dev(noise))
        assert torch.allclose(only_total, total, rtol=1e-6, atol=0)
    with pytest.raises(NotImplementedError):
        s = DDIMScheduler(10)
        DiffusionInferer(s).get_likelihood(_dev(lk["inputs"]), m, s, verbose=False)


def test_controlnet_matches_reference():
    """ControlNet forward (down / mid residuals) and the ControlNet-conditioned inferers against the reference's outputs:
    2-D with class embedding + attention, 3-D with cross-attention, conditioned __call__ and DDIM chain, latent variant with the
    conditioning image resized (nearest) to the latent grid."""
    from generativemodels_amd.inferers import ControlNetDiffusionInferer, ControlNetLatentDiffusionInferer
    from generativemodels_amd.networks.nets import ControlNet
    from generativemodels_amd.networks.schedulers import DDIMScheduler
    fx = load_fixture("controlnet")
    for name, e in fx["forwards"].items():
        m = ControlNet(**e["cfg"]).eval()
        m.load_state_dict(e["state_dict"])
        m = m.to(DEV)
        down, mid = m(_dev(e["x"]), _dev(e["timesteps"]), _dev(e["cond"]), conditioning_scale=e["scale"], context=_dev(e["context"]),
                      class_labels=_dev(e["class_labels"]))
        assert len(down) == len(e["down"])
        for a, b in zip(down, e["down"]):
            _fp32_close(a, b, f"{name} down residual")
        _fp32_close(mid, e["mid"], f"{name} mid residual")
        mb = ControlNet(**e["cfg"]).eval()
        mb.load_state_dict(e["state_dict"])
        mb = mb.to(DEV, torch.bfloat16)
        ctx = None if e["context"] is None else e["context"].bfloat16()
        dn, md = mb(_dev(e["x"].bfloat16()), _dev(e["timesteps"]), _dev(e["cond"].bfloat16()), conditioning_scale=e["scale"], context=_dev(ctx),
                    class_labels=_dev(e["class_labels"]))
        _bf16_close(md, e["mid"], f"{name} bf16 mid residual")
    i = fx["inferer"]
    unet = _nets().DiffusionModelUNet(**i["unet_cfg"]).eval()
    unet.load_state_dict(i["unet_sd"])
    cn = ControlNet(**i["cn_cfg"]).eval()
    cn.load_state_dict(i["cn_sd"])
    unet, cn = unet.to(DEV), cn.to(DEV)
    sch = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sch.set_timesteps(i["steps"])
    inf = ControlNetDiffusionInferer(sch)
    _fp32_close(inf.sample(_dev(i["noise"]), unet, cn, _dev(i["cond"]), sch, verbose=False), i["chain"], "controlnet chain", factor=2.0)
    pred = inf(inputs=_dev(i["call_inputs"]), diffusion_model=unet, controlnet=cn, noise=_dev(i["noise"]), timesteps=_dev(i["call_timesteps"]),
               cn_cond=_dev(i["cond"]))
    _fp32_close(pred, i["call_prediction"], "controlnet inferer __call__")
    l = fx["latent"]
    ae = _nets().AutoencoderKL(**l["ae_cfg"]).eval()
    ae.load_state_dict(l["ae_sd"])
    lunet = _nets().DiffusionModelUNet(**l["unet_cfg"]).eval()
    lunet.load_state_dict(l["unet_sd"])
    lcn = ControlNet(**l["cn_cfg"]).eval()
    lcn.load_state_dict(l["cn_sd"])
    ae, lunet, lcn = ae.to(DEV), lunet.to(DEV), lcn.to(DEV)
    linf = ControlNetLatentDiffusionInferer(sch, scale_factor=l["scale_factor"])
    img = linf.sample(_dev(l["noise"]), ae, lunet, lcn, _dev(l["cond"]), sch, verbose=False)
    _fp32_close(img, l["image"], "controlnet latent chain", factor=2.0)


def test_transformer_and_vqvae_transformer_inferer_match_reference():
    """DecoderOnlyTransformer (full-sequence forward and the KV-cache step API), the causal flag of the attention kernel, the fused
    sampling head and VQVAETransformerInferer (__call__, get_likelihood incl. the sliding window, greedy sample) against the
    reference's outputs / the oracle."""
    from generativemodels_amd import ops
    from generativemodels_amd.inferers import VQVAETransformerInferer
    from generativemodels_amd.networks.nets import DecoderOnlyTransformer
    from generativemodels_amd.utils import Ordering
    fx = load_fixture("transformer")
    for name, e in fx["forwards"].items():
        m = DecoderOnlyTransformer(**e["cfg"]).eval()
        m.load_state_dict(e["state_dict"])
        m = m.to(DEV)
        y = m(_dev(e["tokens"]), context=_dev(e["context"]))
        _fp32_close(y, e["logits"], f"transformer {name}")
        # incremental decoding reproduces the full forward, position by position: native step (one C call per token; models
        # without cross attention) and the op-by-op step issue the same kernels
        for native in (True, False):
            m.native_step = native
            cache = m.new_cache(2, DEV)
            for t in range(e["tokens"].shape[1]):
                lg = m.step(_dev(e["tokens"][:, t:t + 1].contiguous()), t, cache, _dev(e["context"]))
                _fp32_close(lg, e["logits"][:, t], f"transformer {name} step {t} native={native}")
        m.native_step = True
        mb = DecoderOnlyTransformer(**e["cfg"]).eval()
        mb.load_state_dict(e["state_dict"])
        mb = mb.to(DEV, torch.bfloat16)
        ctx = None if e["context"] is None else e["context"].bfloat16()
        _bf16_close(mb(_dev(e["tokens"]), context=_dev(ctx)), e["logits"], f"transformer {name} bf16")
    i = fx["inferer"]
    vq = _nets().VQVAE(**i["vq_cfg"]).eval()
    vq.load_state_dict(i["vq_sd"])
    tr = DecoderOnlyTransformer(**i["tr_cfg"]).eval()
    tr.load_state_dict(i["tr_sd"])
    vq, tr = vq.to(DEV), tr.to(DEV)
    order = Ordering(**i["ordering"])
    inf = VQVAETransformerInferer()
    pred, target, lsd = inf(_dev(i["x"]), vq, tr, order, return_latent=True)
    assert lsd == tuple(i["latent_spatial_dim"]) and torch.equal(target.cpu(), i["target"])
    _fp32_close(pred, i["prediction"], "transformer inferer __call__")
    _fp32_close(inf.get_likelihood(_dev(i["x"]), vq, tr, order), i["likelihood"], "transformer likelihood")
    img = inf.sample((4, 4), torch.full((2, 1), 16, device=DEV), vq, tr, order, top_k=1, verbose=False)
    _fp32_close(img, i["greedy_image"], "greedy sample (KV cache)")
    img_g = VQVAETransformerInferer(use_hip_graph=True).sample((4, 4), torch.full((2, 1), 16, device=DEV), vq, tr, order, top_k=1, verbose=False)
    _fp32_close(img_g, i["greedy_image"], "greedy sample (KV cache, one HIP-graph replay per token, device-side position)")
    # the sampling head against the oracle on the teacher-forced logits: temperature / top-k variants incl. ties and k >= V
    logits = i["logits"].reshape(-1, i["logits"].shape[-1])
    for temp, k in [(1.0, None), (0.7, 5), (1.3, 1), (1.0, 17), (2.0, 40)]:
        want = R.transformer_sample_probs(logits.clone(), temp, k, 16)
        got = ops.sample_probs(_dev(logits), temp, k, 16).cpu()
        assert (got - want).abs().max().item() <= 1e-6, (temp, k)
        assert torch.equal(got == 0, want == 0)
    tied = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0, -1.0]])
    assert torch.equal(ops.sample_probs(_dev(tied), 1.0, 2, 5).cpu() > 0, R.transformer_sample_probs(tied.clone(), 1.0, 2, 5) > 0)
    w = fx["window"]
    tr2 = DecoderOnlyTransformer(**w["tr_cfg"]).eval()
    tr2.load_state_dict(w["tr_sd"])
    tr2 = tr2.to(DEV)
    o2 = Ordering("raster_scan", 2, (1, 2, 2))
    _fp32_close(inf.get_likelihood(_dev(w["x"]), vq, tr2, o2), w["likelihood"], "windowed likelihood")
    s = inf.sample((2, 2), torch.full((2, 1), 16, device=DEV), vq, tr2, o2, top_k=3, verbose=False)  # window slides: recompute path
    assert s.shape == (2, 1, 8, 8) and torch.isfinite(s).all()


def _grid_sd(e):
    return R.synthetic_state_dict(e["shapes"], seed=e["seed"])


def test_constructor_argument_grid_matches_reference():
    """DiffusionModelUNet / AutoencoderKL / VQVAE over the constructor-argument grid of tests/golden/config_grid.pt (16 + 10 + 6
    configurations in the spirit of the reference's own shape tests), fp32, against the reference's outputs."""
    fx = load_fixture("config_grid")
    for e in fx["unet"]:
        m = _nets().DiffusionModelUNet(**e["cfg"]).eval()
        m.load_state_dict(_grid_sd(e), strict=True)
        m = m.to(DEV)
        y = m(_dev(e["x"]), _dev(e["timesteps"]), context=_dev(e["context"]), class_labels=_dev(e["class_labels"]))
        _fp32_close(y, e["y"], f"unet {e['cfg']}")
    for e in fx["aekl"]:
        m = _nets().AutoencoderKL(**e["cfg"]).eval()
        m.load_state_dict(_grid_sd(e), strict=True)
        m = m.to(DEV)
        mu, sigma = m.encode(_dev(e["x"]))
        _fp32_close(mu, e["z_mu"], f"aekl mu {e['cfg']}")
        _fp32_close(sigma, e["z_sigma"], f"aekl sigma {e['cfg']}")
        _fp32_close(m.decode(_dev(e["z_mu"])), e["reconstruction"], f"aekl decode {e['cfg']}")
    for e in fx["vqvae"]:
        m = _nets().VQVAE(**e["cfg"]).eval()
        m.load_state_dict(_grid_sd(e), strict=True)
        m = m.to(DEV)
        _fp32_close(m.encode(_dev(e["x"])), e["z"], f"vqvae z {e['cfg']}")
        _fp32_close(m.decode_samples(_dev(e["indices"])), e["reconstruction"], f"vqvae decode {e['cfg']}")
        idx = m.index_quantize(_dev(e["x"])).cpu()
        # indices are an argmin over fp32 distances: a near-tie may flip under a different summation order; demand >= 99 % agreement
        assert (idx == e["indices"]).float().mean().item() >= 0.99, f"vqvae indices {e['cfg']}"


@pytest.mark.parametrize("shape", [(1, 1, 20, 24, 28), (2, 1, 12, 36, 20)])
def test_unet3d_ragged_volume_matches_oracle(shape):
    """Volumes whose extents are not multiples of the 4x4x16 / 2x4x16 tiles of the LDS-DMA kernels (partial tiles on every axis, at
    every resolution level, through the stride-2 and up-sampling convolutions and the fused shortcut), fp32, against the CPU oracle."""
    cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(32, 64, 64), attention_levels=(False, False, True),
               num_res_blocks=(1, 2, 1), num_head_channels=(0, 0, 32), norm_num_groups=16)
    torch.manual_seed(0)
    m = _nets().DiffusionModelUNet(**cfg).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    R.derandomize_zeros(sd)
    m.load_state_dict(sd)
    x = torch.randn(shape, generator=torch.Generator().manual_seed(3))
    t = torch.tensor([700, 40][:shape[0]])
    with torch.no_grad():
        want = R.unet_forward(sd, cfg, x, t)
    got = m.to(DEV)(_dev(x), _dev(t))
    _fp32_close(got, want, f"ragged unet {shape}")
    yb = m.to(DEV, torch.bfloat16)(_dev(x.bfloat16()), _dev(t))
    _bf16_close(yb, want, f"ragged unet bf16 {shape}")


SWEEP = [
    # (name, cfg, input shape, context shape): geometries on either side of the round-6 kernel policies -- token rows below / above 2 048 and 8 192 (wide token GEMM, 2 / 4
    # blocks per wave), statistic tables of 1 ... 128 rows (GroupNorm finalised in the consumer), C_in <= 4 image inputs (edge kernel), narrow output heads, ragged extents
    ("2d-configs0-batch16", dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1,
                                 num_head_channels=64), (16, 1, 64, 64), None),
    ("2d-ragged-3levels", dict(spatial_dims=2, in_channels=3, out_channels=2, num_channels=(64, 128, 128), attention_levels=(False, True, True), num_res_blocks=2,
                               num_head_channels=(0, 64, 32)), (3, 3, 40, 56), None),
    ("2d-attention-everywhere", dict(spatial_dims=2, in_channels=2, out_channels=4, num_channels=(32, 32, 64), attention_levels=(True, True, True), num_res_blocks=1,
                                     num_head_channels=32), (9, 2, 32, 32), None),
    ("3d-latent-like", dict(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1,
                            num_head_channels=32), (2, 4, 16, 16, 16), None),
    ("3d-ragged", dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(32, 64, 128), attention_levels=(False, False, True), num_res_blocks=(1, 1, 2),
                       num_head_channels=(0, 0, 64)), (1, 1, 24, 20, 36), None),
    ("2d-cross-attention", dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1,
                                num_head_channels=(0, 32), with_conditioning=True, cross_attention_dim=24, transformer_num_layers=1), (2, 1, 32, 32), (2, 5, 24)),
]


@pytest.mark.parametrize("case", SWEEP, ids=lambda c: c[0])
def test_unet_configuration_sweep_against_the_oracle(case):
    """(round 6) DiffusionModelUNet.forward under no_grad -- the path DiffusionInferer.sample takes -- over configurations chosen to sit on either side of the kernel
    policies this round added (SWEEP above), fp32 and bf16, against the CPU oracle's restatement of the reference forward (diffusion_model_unet.py:1869-1943)."""
    name, cfg, shape, ctx_shape = case
    torch.manual_seed(11)
    m = _nets().DiffusionModelUNet(**cfg).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    R.derandomize_zeros(sd)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g)
    t = torch.randint(0, 1000, (shape[0],), generator=g)
    ctx = None if ctx_shape is None else torch.randn(ctx_shape, generator=g)
    with torch.no_grad():
        want = R.unet_forward(sd, cfg, x, t, ctx)
        got = m.to(DEV)(_dev(x), _dev(t), context=_dev(ctx))
        _fp32_close(got, want, f"sweep {name} fp32", factor=2.0)
        yb = m.to(DEV, torch.bfloat16)(_dev(x.bfloat16()), _dev(t), context=None if ctx is None else _dev(ctx.bfloat16()))
        _bf16_close(yb, want, f"sweep {name} bf16")
        again = m(_dev(x.bfloat16()), _dev(t), context=None if ctx is None else _dev(ctx.bfloat16()))
    assert torch.equal(again, yb)  # run-to-run bitwise


def test_spade_networks_match_reference():
    """SPADE block, SPADEDiffusionModelUNet (2-D with attention and a coarser segmentation; 3-D with cross-attention + resblock_updown),
    SPADEAutoencoderKL encode / decode and the seg-conditioned latent DDIM chain against the reference's outputs (fp32), plus bf16
    closeness; the per-layer (1 + gamma, beta) maps are cached across the chain's steps."""
    from generativemodels_amd.inferers import LatentDiffusionInferer
    from generativemodels_amd.networks.blocks import SPADE
    from generativemodels_amd.networks.nets import SPADEAutoencoderKL, SPADEDiffusionModelUNet
    from generativemodels_amd.networks.schedulers import DDIMScheduler
    fx = load_fixture("spade")
    for name, e in fx["blocks"].items():
        m = SPADE(**e["kwargs"]).eval()
        m.load_state_dict(e["state_dict"])
        m = m.to(DEV)
        _fp32_close(m(_dev(e["x"]), _dev(e["seg"])), e["y"], f"spade block {name}", factor=2.0)
    for name, e in fx["unets"].items():
        m = SPADEDiffusionModelUNet(**e["cfg"]).eval()
        m.load_state_dict(e["state_dict"])
        m = m.to(DEV)
        seg = _dev(e["seg"])
        y = m(_dev(e["x"]), _dev(e["timesteps"]), seg, context=_dev(e["context"]))
        _fp32_close(y, e["y"], name)
        _fp32_close(m(_dev(e["x"]), _dev(e["timesteps"]), seg, context=_dev(e["context"])), e["y"], f"{name} (cached maps)")
        mb = SPADEDiffusionModelUNet(**e["cfg"]).eval()
        mb.load_state_dict(e["state_dict"])
        mb = mb.to(DEV, torch.bfloat16)
        ctx = None if e["context"] is None else e["context"].bfloat16()
        yb = mb(_dev(e["x"].bfloat16()), _dev(e["timesteps"]), _dev(e["seg"].bfloat16()), context=_dev(ctx))
        _bf16_close(yb, e["y"], f"{name} bf16")
    for name, e in fx["aekls"].items():
        m = SPADEAutoencoderKL(**e["cfg"]).eval()
        m.load_state_dict(e["state_dict"])
        m = m.to(DEV)
        z_mu, z_sigma = m.encode(_dev(e["x"]))
        _fp32_close(z_mu, e["z_mu"], f"{name} z_mu")
        _fp32_close(m.decode(_dev(e["z_mu"]), _dev(e["seg"])), e["decoded"], f"{name} decode", factor=2.0)
    l = fx["latent"]
    ae = SPADEAutoencoderKL(**l["ae_cfg"]).eval()
    ae.load_state_dict(l["ae_sd"])
    unet = SPADEDiffusionModelUNet(**l["unet_cfg"]).eval()
    unet.load_state_dict(l["unet_sd"])
    ae, unet = ae.to(DEV), unet.to(DEV)
    sch = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sch.set_timesteps(l["steps"])
    inf = LatentDiffusionInferer(sch, scale_factor=l["scale_factor"])
    img = inf.sample(_dev(l["noise"]), ae, unet, sch, verbose=False, seg=_dev(l["seg"]))
    _fp32_close(img, l["image"], "spade latent chain", factor=4.0)
