"""CPU (-m "not gpu"): host-side logic of the drop-in layer -- constructor contracts, state_dict key parity with the
reference (via the committed golden fixtures), scheduler tables / timesteps (bit-exact vs the reference goldens), the
ComponentStore plugin point, and the fail-loudly behaviour without a GPU. No kernel is launched here."""
import pytest
import torch

from _util import load_fixture
from generativemodels_amd.inferers import DiffusionInferer, LatentDiffusionInferer
from generativemodels_amd.networks.nets import VQVAE, AutoencoderKL, DiffusionModelUNet
from generativemodels_amd.networks.schedulers import DDIMScheduler, DDPMScheduler, NoiseSchedules
from generativemodels_amd.utils import ComponentStore, unsqueeze_left, unsqueeze_right

KINDS = {"unet": DiffusionModelUNet, "aekl": AutoencoderKL, "vqvae": VQVAE}
FIXTURES = ["unet2d_c1a", "unet3d_c1a", "unet2d_c1b", "unet3d_c2mini", "unet2d_cond", "unet3d_cond", "aekl2d", "aekl3d_brainlike",
            "aekl3d_convT", "vqvae3d", "vqvae2d_odd"]


@pytest.mark.parametrize("name", FIXTURES)
def test_state_dict_keys_and_shapes_match_reference(name):
    fx = load_fixture(name)
    m = KINDS[fx["kind"]](**fx["cfg"])
    ours = m.state_dict()
    ref = fx["state_dict"]
    assert set(ours.keys()) == set(ref.keys())
    for k in ref:
        assert tuple(ours[k].shape) == tuple(ref[k].shape), k
    m.load_state_dict(ref, strict=True)


def test_fresh_unet_has_the_reference_zero_initialised_layers():
    m = DiffusionModelUNet(2, 1, 1, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1, norm_num_groups=8,
                           with_conditioning=True, cross_attention_dim=4)
    sd = m.state_dict()
    zeros = [k for k, v in sd.items() if v.is_floating_point() and v.abs().max() == 0]
    assert "out.2.conv.weight" in zeros and "down_blocks.0.resnets.0.conv2.conv.weight" in zeros
    assert "down_blocks.1.attentions.0.proj_out.conv.weight" in zeros
    assert "down_blocks.0.resnets.0.conv1.conv.weight" not in zeros


def test_unet_constructor_errors():
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, with_conditioning=True, cross_attention_dim=None)
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, with_conditioning=False, cross_attention_dim=3)
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, num_channels=(8, 12), attention_levels=(False, False), num_res_blocks=1, norm_num_groups=8)
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, num_channels=(8, 8, 8), attention_levels=(False, False), num_res_blocks=1, norm_num_groups=8)
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, num_channels=(8, 8), attention_levels=(False, False), num_res_blocks=(1, 1, 1), norm_num_groups=8)
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, num_channels=(8, 8), attention_levels=(False, True), num_res_blocks=1, norm_num_groups=8,
                           num_head_channels=(4, 4, 4))
    with pytest.raises(ValueError):
        DiffusionModelUNet(2, 1, 1, num_channels=(8, 8), attention_levels=(False, False), num_res_blocks=1, norm_num_groups=8,
                           dropout_cattn=3.0)
    with pytest.raises(ValueError):  # default 4-tuples do not fit a 2-level model
        DiffusionModelUNet(2, 1, 1, num_channels=(8, 8), attention_levels=(False, False), norm_num_groups=8)


def test_autoencoder_and_vqvae_constructor_errors():
    with pytest.raises(ValueError):
        AutoencoderKL(2, num_channels=(24, 24, 24), attention_levels=(False, False, False), num_res_blocks=1, norm_num_groups=16)
    with pytest.raises(ValueError):
        AutoencoderKL(2, num_channels=(8, 8, 8), attention_levels=(False, False), num_res_blocks=1, norm_num_groups=8)
    with pytest.raises(ValueError):
        AutoencoderKL(2, num_channels=(8, 8, 8), attention_levels=(False, False, False), num_res_blocks=(1, 1), norm_num_groups=8)
    with pytest.raises(ValueError):
        VQVAE(2, 1, 1, num_channels=(8, 16), num_res_channels=(8, 16, 16), downsample_parameters=((2, 4, 1, 1),) * 2,
              upsample_parameters=((2, 4, 1, 1, 0),) * 2)
    with pytest.raises(ValueError):
        VQVAE(2, 1, 1, num_channels=(8, 16), num_res_channels=(8, 16), downsample_parameters=((2, 4, 1),) * 2,
              upsample_parameters=((2, 4, 1, 1, 0),) * 2)
    with pytest.raises(ValueError):
        VQVAE(2, 1, 1, num_channels=(8, 16), num_res_channels=(8, 16), downsample_parameters=((2, 4, 1, 1),) * 2,
              upsample_parameters=((2, 4, 1, 1),) * 2)
    with pytest.raises(ValueError):
        VQVAE(2, 1, 1, num_channels=(8, 16), num_res_channels=(8, 16), downsample_parameters=((2, 4, 1, 1),) * 3,
              upsample_parameters=((2, 4, 1, 1, 0),) * 2)
    v = VQVAE(3, 1, 1, num_channels=(8, 16), num_res_channels=8, downsample_parameters=(2, 4, 1, 1), upsample_parameters=(2, 4, 1, 1, 0),
              num_res_layers=1, num_embeddings=16, embedding_dim=8)
    assert "quantizer.quantizer.ema_w" in v.state_dict()


def test_scheduler_tables_timesteps_and_errors():
    fx = load_fixture("schedulers")
    for sname, e in fx["tables"].items():
        d = DDIMScheduler(1000, schedule=sname, **e["kw"])
        assert torch.equal(d.betas, e["betas"]) and torch.equal(d.alphas, e["alphas"]) and torch.equal(d.alphas_cumprod, e["alphas_cumprod"])
        d.set_timesteps(50)
        assert torch.equal(d.timesteps, e["timesteps50"])
        p = DDPMScheduler(1000, schedule=sname, **e["kw"])
        assert torch.equal(p.alphas_cumprod, e["alphas_cumprod"])
    d = DDIMScheduler(1000, steps_offset=1)
    d.set_timesteps(100)
    assert len(d.timesteps) == 100 and int(d.timesteps[-1]) == 1
    p = DDPMScheduler(1000)
    p.set_timesteps(100)
    assert len(p.timesteps) == 100 and int(p.timesteps[0]) == 990
    for cls in (DDIMScheduler, DDPMScheduler):
        with pytest.raises(ValueError):
            cls(10).set_timesteps(11)
        with pytest.raises(ValueError):
            cls(prediction_type="nope")
        with pytest.raises(ValueError):
            cls(clip_sample_min=1, clip_sample_max=1)
    with pytest.raises(ValueError):
        DDPMScheduler(variance_type="nope")
    with pytest.raises(ValueError):
        DDIMScheduler(schedule="not_registered")
    assert float(DDIMScheduler(set_alpha_to_one=False).final_alpha_cumprod) == float(DDIMScheduler().alphas_cumprod[0])


def test_pndm_tables_state_and_errors():
    """PNDMScheduler host logic (reference pndm.py:80-163 and the shape pins of tests/test_scheduler_pndm.py): timestep tables,
    the evaluation count (100 requested -> 109), constructor / set_timesteps errors, no CPU fallback."""
    from generativemodels_amd.networks.schedulers import PNDMScheduler

    fx = load_fixture("pndm_likelihood")
    for (n, skip), e in fx["tables"].items():
        s = PNDMScheduler(1000, skip_prk_steps=skip)
        s.set_timesteps(n)
        assert torch.equal(s.timesteps, e["timesteps"]) and s.num_inference_steps == e["num_inference_steps"], (n, skip)
        assert len(s.prk_timesteps) == len(e["prk"]) and torch.equal(torch.as_tensor(s.plms_timesteps.copy()), e["plms"])
        assert s.counter == 0 and s.ets == []
    s = PNDMScheduler(1000)
    s.set_timesteps(100)
    assert s.num_inference_steps == 109 and len(s.timesteps) == 109
    assert float(PNDMScheduler().final_alpha_cumprod) == float(PNDMScheduler().alphas_cumprod[0])
    assert float(PNDMScheduler(set_alpha_to_one=True).final_alpha_cumprod) == 1.0
    with pytest.raises(ValueError):
        PNDMScheduler(10).set_timesteps(11)
    with pytest.raises(ValueError):
        PNDMScheduler(prediction_type="sample")  # PNDM knows epsilon and v_prediction only (pndm.py:43-53)
    x = torch.zeros(1, 1, 8, 8)
    with pytest.raises(RuntimeError, match="MI355X"):
        PNDMScheduler(10).step(x, 5, x)
    with pytest.raises(NotImplementedError):
        DiffusionInferer(PNDMScheduler(10)).get_likelihood(x, lambda *a, **k: x, verbose=False)  # DDPM only (inferer.py:176-180)


def test_user_registered_noise_schedule_feeds_the_scheduler():
    name = "halfway_test_schedule"
    if name not in NoiseSchedules:
        @NoiseSchedules.add_def(name, "test schedule")
        def _sched(num_train_timesteps, level=0.5):
            """constant beta"""
            return torch.full((num_train_timesteps,), level)
    s = DDPMScheduler(20, schedule=name, level=0.25)
    assert torch.allclose(s.alphas_cumprod, torch.cumprod(torch.full((20,), 0.75), 0))


def test_component_store_contract():
    st = ComponentStore("T", "desc")
    st.add("a", "first", 1)
    with pytest.raises(ValueError):
        st.add("not valid!", "x", 2)
    with pytest.raises(ValueError):
        st["missing"]
    assert "a" in st and len(st) == 1 and st.a == 1 and list(st) == [("a", 1)] and "Component Store 'T'" in str(st)
    x = torch.zeros(3)
    assert unsqueeze_right(x, 3).shape == (3, 1, 1) and unsqueeze_left(x, 3).shape == (1, 1, 3)


def test_inferer_argument_checks_and_no_cpu_fallback():
    s = DDIMScheduler(10)
    with pytest.raises(ValueError):
        LatentDiffusionInferer(s, ldm_latent_shape=[4, 4], autoencoder_latent_shape=None)
    inf = DiffusionInferer(s)
    x = torch.zeros(1, 1, 8, 8)
    with pytest.raises(NotImplementedError):
        inf.sample(x, lambda *a, **k: x, s, mode="bad", verbose=False)
    m = DiffusionModelUNet(2, 1, 1, num_channels=[8], norm_num_groups=8, attention_levels=[True], num_res_blocks=1, num_head_channels=8)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(x, torch.tensor([1]))  # CPU tensors are refused: there is no eager fallback
    with pytest.raises(RuntimeError, match="MI355X"):
        s.step(x, 5, x)
    with pytest.raises(RuntimeError, match="MI355X"):
        s.add_noise(x, x, torch.tensor([1]))


def test_captured_forward_cache_keys_on_shapes_and_parameter_versions(monkeypatch):
    """DiffusionInferer._cached_graph (host logic, no GPU: the capture itself is replaced by a recorder): one capture per (model, shapes, dtype,
    conditioning shape) reused across calls; an in-place parameter update (its `_version` moves) or a train() / eval() switch re-captures and
    drops the stale entry; at most GRAPH_CACHE_SIZE captures are kept."""
    from generativemodels_amd.inferers import inferer as I

    built = []

    class Recorder:
        signature = staticmethod(I._GraphedUNet.signature)

        def __init__(self, model, x, t, ctx, row=None):
            built.append((id(model), tuple(x.shape), None if ctx is None else tuple(ctx.shape)))

    monkeypatch.setattr(I, "_GraphedUNet", Recorder)
    inf = DiffusionInferer(DDIMScheduler(10), use_hip_graph=True)
    m1, m2 = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    x, t, ctx = torch.zeros(1, 4, 8, 8), torch.zeros(1), torch.zeros(1, 3, 4)
    g = inf._cached_graph(m1, x, t, None)
    assert inf._cached_graph(m1, x.clone(), t, None) is g and len(built) == 1          # same shapes: the same capture
    g_ctx = inf._cached_graph(m1, x, t, ctx)
    assert g_ctx is not g and len(built) == 2                                          # conditioning changes the graph
    assert inf._cached_graph(m1, x, t, None) is g and inf._cached_graph(m1, x, t, ctx) is g_ctx
    with torch.no_grad():
        m1.weight.mul_(2.0)                                                             # an optimizer step
    g2 = inf._cached_graph(m1, x, t, None)
    assert g2 is not g and len(built) == 3
    assert all(not (e[0]() is m1 and e[3] is g) for e in inf._graph_cache)             # the stale capture is gone
    m1.train(False)
    m1.train(True)
    assert inf._cached_graph(m1, x, t, None) is g2                                      # mode unchanged in the end: still valid
    m1.eval()
    assert inf._cached_graph(m1, x, t, None) is not g2                                  # eval() vs train(): another forward
    inf._cached_graph(m2, x, t, None)
    inf._cached_graph(m2, torch.zeros(2, 4, 8, 8), t, None)
    assert len(inf._graph_cache) <= inf.GRAPH_CACHE_SIZE
    # writes through .data do not bump _version (an EMA-weight swap, module.to(), .half()): the storage / dtype are part of the signature
    inf.clear_graph_cache()
    m3 = torch.nn.Linear(4, 4).eval()
    g3 = inf._cached_graph(m3, x, t, None)
    v0 = m3.weight._version
    m3.weight.data = torch.ones(4, 4)                                                  # p.data = ema_p.data
    assert m3.weight._version == v0                                                    # (what ADVICE r4 pointed at)
    g4 = inf._cached_graph(m3, x, t, None)
    assert g4 is not g3
    m3.to(torch.float64)
    g5 = inf._cached_graph(m3, x, t, None)
    assert g5 is not g4
    # the autocast compute dtype is part of the signature too
    import generativemodels_amd as gm
    with gm.autocast(torch.bfloat16):
        g6 = inf._cached_graph(m3, x, t, None)
    assert g6 is not g5
    # a dropped model releases its capture (the cached object holds no strong reference to it)
    inf.clear_graph_cache()
    import gc
    import weakref
    m4 = torch.nn.Linear(4, 4)
    inf._cached_graph(m4, x, t, None)
    w = weakref.ref(m4)
    del m4
    gc.collect()
    assert w() is None
    inf._cached_graph(m2, x, t, None)                                                  # the next insertion prunes the dead entry
    assert all(e[0]() is not None for e in inf._graph_cache)


def test_install_as_generative_aliases_the_reference_import_paths():
    import subprocess
    import sys
    code = ("import generativemodels_amd as g; g.install_as_generative(); "
            "from generative.networks.nets import DiffusionModelUNet, AutoencoderKL, VQVAE; "
            "from generative.networks.schedulers import DDIMScheduler, DDPMScheduler; "
            "from generative.inferers import DiffusionInferer, LatentDiffusionInferer; "
            "from generative.networks.layers import VectorQuantizer; "
            "from generative.networks.blocks.spade_norm import SPADE; from generative.networks.nets import SPADEDiffusionModelUNet; print('ok')")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_controlnet_state_dict_and_constructor_errors():
    """ControlNet host logic (reference controlnet.py:190-365): strict state_dict compatibility incl. the bare first zero conv,
    zero-initialised control convolutions, constructor errors, copy_weights_to_controlnet, no CPU fallback."""
    from generativemodels_amd.networks.nets import ControlNet, copy_weights_to_controlnet

    fx = load_fixture("controlnet")
    for name, e in fx["forwards"].items():
        m = ControlNet(**e["cfg"])
        assert set(m.state_dict()) == set(e["state_dict"]), name
        for k, v in m.state_dict().items():
            assert tuple(v.shape) == tuple(e["state_dict"][k].shape), (name, k)
        m.load_state_dict(e["state_dict"], strict=True)
    m = ControlNet(2, 1, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1, norm_num_groups=8, num_head_channels=8)
    assert "controlnet_down_blocks.0.weight" in m.state_dict() and "controlnet_down_blocks.1.conv.weight" in m.state_dict()
    for k, v in m.state_dict().items():
        if k.startswith("controlnet_down_blocks") or k.startswith("controlnet_mid_block") or k.startswith("controlnet_cond_embedding.conv_out"):
            assert float(v.abs().max()) == 0.0, k
    unet = DiffusionModelUNet(2, 1, 1, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1, norm_num_groups=8, num_head_channels=8)
    copy_weights_to_controlnet(m, unet, verbose=False)
    assert torch.equal(m.down_blocks[0].resnets[0].conv1.conv.weight, unet.down_blocks[0].resnets[0].conv1.conv.weight)
    with pytest.raises(ValueError):
        ControlNet(2, 1, with_conditioning=True)
    with pytest.raises(ValueError):
        ControlNet(2, 1, cross_attention_dim=3)
    with pytest.raises(ValueError):
        ControlNet(2, 1, num_channels=(8, 12), norm_num_groups=8, attention_levels=(False, False))
    with pytest.raises(ValueError):
        ControlNet(2, 1, num_channels=(8, 8), norm_num_groups=8, attention_levels=(False,))
    with pytest.raises(ValueError):
        ControlNet(2, 1, num_channels=(8, 8), norm_num_groups=8, attention_levels=(False, False), num_res_blocks=(1, 1, 1))
    x = torch.zeros(1, 1, 8, 8)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(x, torch.tensor([1]), x)


def test_spade_state_dicts_constructor_errors_and_no_cpu_fallback():
    """SPADE host logic (reference spade_norm.py:33-77, spade_diffusion_model_unet.py:641-834, spade_autoencoderkl.py:315-408): strict
    state_dict compatibility with the reference's keys (incl. `param_free_norm.N.*` only where the GroupNorm is affine), constructor
    errors under the reference's class names, forward argument checks, no CPU fallback."""
    from generativemodels_amd.networks.blocks import SPADE
    from generativemodels_amd.networks.nets import SPADEAutoencoderKL, SPADEDiffusionModelUNet

    fx = load_fixture("spade")
    for name, e in fx["blocks"].items():
        m = SPADE(**e["kwargs"])
        assert set(m.state_dict()) == set(e["state_dict"]), name
        m.load_state_dict(e["state_dict"], strict=True)
    for group, cls in (("unets", SPADEDiffusionModelUNet), ("aekls", SPADEAutoencoderKL)):
        for name, e in fx[group].items():
            m = cls(**e["cfg"])
            assert set(m.state_dict()) == set(e["state_dict"]), name
            for k, v in m.state_dict().items():
                assert tuple(v.shape) == tuple(e["state_dict"][k].shape), (name, k)
            m.load_state_dict(e["state_dict"], strict=True)
    u = SPADEDiffusionModelUNet(2, 1, 1, label_nc=3, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1, norm_num_groups=8,
                                num_head_channels=8, spade_intermediate_channels=16)
    assert any(k.endswith("norm1.param_free_norm.N.weight") for k in u.state_dict())          # affine GroupNorm inside the UNet's SPADE
    assert not any("down_blocks" in k and "mlp_shared" in k for k in u.state_dict())           # encoder blocks are the plain ones
    a = SPADEAutoencoderKL(2, label_nc=3, num_channels=(8, 16), attention_levels=(False, False), num_res_blocks=1, norm_num_groups=8,
                           latent_channels=4, with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False, spade_intermediate_channels=16)
    assert not any("param_free_norm" in k for k in a.state_dict())                             # affine-free GroupNorm: no parameters
    assert a.decoder.blocks[1].norm1.eps == 1e-5                                               # nn.GroupNorm default, not norm_eps
    with pytest.raises(ValueError, match="SPADEDiffusionModelUNet"):
        SPADEDiffusionModelUNet(2, 1, 1, label_nc=3, with_conditioning=True)
    with pytest.raises(ValueError, match="SPADEAutoencoderKL"):
        SPADEAutoencoderKL(2, label_nc=3, num_channels=(8, 12), norm_num_groups=8, attention_levels=(False, False))
    with pytest.raises(NotImplementedError):
        SPADE(3, 8, norm="BATCH")
    x = torch.zeros(1, 1, 8, 8)
    with pytest.raises(ValueError):
        u(x, torch.tensor([1]), None)
    with pytest.raises(ValueError):
        u(x, torch.tensor([1]), torch.zeros(1, 2, 8, 8))   # label_nc mismatch
    with pytest.raises(RuntimeError, match="MI355X"):
        u(x, torch.tensor([1]), torch.zeros(1, 3, 8, 8))
    assert u.supports_training()  # round 3: the SPADE variant trains natively too (forward_train takes `seg`)
    with pytest.raises(ValueError):
        u.forward_train(x, torch.tensor([1]))   # ... and refuses to run without its segmentation


def test_transformer_state_dict_ordering_and_errors():
    """DecoderOnlyTransformer / SABlock / Ordering host logic: reference state_dict keys (incl. the causal_mask buffer key, emitted on
    save and ignored on load), ordering permutations, constructor errors, no CPU fallback."""
    import numpy as np

    from generativemodels_amd.networks.blocks import SABlock, TransformerBlock
    from generativemodels_amd.networks.nets import DecoderOnlyTransformer
    from generativemodels_amd.utils import Ordering

    fx = load_fixture("transformer")
    for name, e in fx["forwards"].items():
        m = DecoderOnlyTransformer(**e["cfg"])
        sd = m.state_dict()
        assert set(sd) == set(e["state_dict"]), name
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(e["state_dict"][k].shape), (name, k)
        assert torch.equal(sd["blocks.0.attn.causal_mask"], e["state_dict"]["blocks.0.attn.causal_mask"])
        m.load_state_dict(e["state_dict"], strict=True)
    for key, e in fx["orderings"].items():
        o = Ordering(**e["kw"])
        assert np.array_equal(o.get_sequence_ordering(), e["order"].numpy()) and np.array_equal(o.get_revert_sequence_ordering(), e["revert"].numpy())
        x = torch.arange(len(e["order"]))
        assert torch.equal(o(x)[torch.as_tensor(o.get_revert_sequence_ordering().copy())], x)
    np.random.seed(0)
    r = Ordering("random", 2, (1, 3, 3))
    assert sorted(r.get_sequence_ordering().tolist()) == list(range(9))
    with pytest.raises(ValueError):
        Ordering("hilbert", 2, (1, 2, 2))
    with pytest.raises(ValueError):
        Ordering("raster_scan", 2, (1, 2, 2, 2))
    with pytest.raises(ValueError):
        Ordering("raster_scan", 2, (1, 2, 2), transformation_order=("reflect", "reflect"))
    with pytest.raises(ValueError):
        Ordering("raster_scan", 2, (1, 2, 2), transformation_order=("shear",))
    with pytest.raises(ValueError):
        SABlock(16, 3)
    with pytest.raises(ValueError):
        SABlock(16, 4, causal=True)
    with pytest.raises(ValueError):
        SABlock(16, 4, dropout_rate=1.5)
    with pytest.raises(ValueError):
        TransformerBlock(16, 64, 3)
    m = DecoderOnlyTransformer(num_tokens=10, max_seq_len=8, attn_layers_dim=16, attn_layers_depth=1, attn_layers_heads=2)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros((1, 4), dtype=torch.long))


def test_stride2_subpixel_tap_tables_reproduce_a_transposed_convolution():
    """ops.stride2_subpixel_taps (the per-parity 2-tap kernels behind the stride-2 data gradient and the VQ-VAE / AutoencoderKL transposed
    convolutions) against a brute-force 1-D transposed convolution: out[2i + parity] = t0 * x[i - 1 + parity] + t1 * x[i + parity]."""
    import numpy as np

    from generativemodels_amd import ops

    rng = np.random.default_rng(0)
    for K, pad, opad in ((3, 1, 1), (3, 0, 0), (4, 1, 0)):
        n = 7
        x, w = rng.standard_normal(n), rng.standard_normal(K)
        full = np.zeros(2 * n + K + 2)
        for o in range(n):
            for k in range(K):
                full[2 * o + k] += x[o] * w[k]           # out[u + pad] with u = 2 o - pad + k
        want = full[pad:pad + 2 * n]
        assert (n - 1) * 2 - pad - 1 + K + opad == 2 * n  # (padding high = 1: the geometries the sub-pixel path accepts)
        taps = ops.stride2_subpixel_taps(K, pad)
        xp = np.concatenate([[0.0], x, [0.0]])            # zero rows outside the volume
        got = np.zeros(2 * n)
        for i in range(n):
            for par in (0, 1):
                t0, t1 = taps[par]
                got[2 * i + par] = (0.0 if t0 is None else w[t0] * xp[i + par]) + (0.0 if t1 is None else w[t1] * xp[i + 1 + par])
        assert np.allclose(got, want), (K, pad)
        assert ops.stride2_subpixel_covers(K, pad)
    # K = 4 with padding 0 also doubles the extent (padding high 2), but for odd outputs tap W[3] belongs to input i - 1, which the
    # sub-pixel kernel's parity 1 never reads: the geometry must be refused (it takes the generic transposed path), not mis-computed
    assert not ops.stride2_subpixel_covers(4, 0)
    n, K, pad = 7, 4, 0
    x, w = rng.standard_normal(n), rng.standard_normal(K)
    full = np.zeros(2 * n + K + 2)
    for o in range(n):
        for k in range(K):
            full[2 * o + k] += x[o] * w[k]
    taps = ops.stride2_subpixel_taps(K, pad)
    xp = np.concatenate([[0.0], x, [0.0]])
    got = np.array([(0.0 if taps[u & 1][0] is None else w[taps[u & 1][0]] * xp[(u >> 1) + (u & 1)])
                    + (0.0 if taps[u & 1][1] is None else w[taps[u & 1][1]] * xp[(u >> 1) + 1 + (u & 1)]) for u in range(2 * n)])
    assert not np.allclose(got, full[:2 * n])   # (what the two-tap form would have produced: wrong -- hence the guard)
    with pytest.raises(ValueError):
        ops.packed_stride2_dgrad_weight(torch.zeros(8, 8, 4, 4, 4), torch.float32, 0)
    assert not ops.stride2_subpixel_covers(5, 1) and not ops.stride2_subpixel_covers(3, 2)


def test_tile_configuration_policy_of_the_lds_dma_convolutions():
    """ops._choose_conv_cfg on descriptors only (no GPU): large prologue-free stride-1 3x3x3 convolutions take the 4-wave x 64-voxel tile
    (cfg 14, the two-operand-set tap loop), a fused GroupNorm prologue or a grid below one wave of work-groups keeps cfg 11 (its
    fused-prologue instantiation / split-K form), stride 2 takes cfg 15; GM_CONV_WIDE_WAVES=0 restores cfg 11 everywhere."""
    from generativemodels_amd import _native as nat
    from generativemodels_amd import ops

    def desc(size, cin, cout, stride=1, pre=False):
        d = nat.GmConvDesc()
        out = size // stride
        vals = dict(N=1, Cin=cin, Cout=cout, Ds=size, Hs=size, Ws=size, Do=out, Ho=out, Wo=out, kd=3, kh=3, kw=3, sd=stride, sh=stride, sw=stride,
                    pd=1, ph=1, pw=1, dd=1, dh=1, dw=1, in_mode=0, fd=1, fh=1, fw=1, dtype=1, x_ld=cin, y_ld=cout)
        for k, v in vals.items():
            setattr(d, k, v)
        d.x, d.y, d.w = 0x1000, 0x2000, 0x3000
        if pre:
            d.pre_scale, d.pre_shift, d.pre_act = 0x4000, 0x5000, 1
        return d

    def chosen(size, cin, cout, **kw):
        d = desc(size, cin, cout, **kw)
        ops._choose_conv_cfg(d, (size // kw.get("stride", 1)) ** 3, only=ops.DMA_CFGS)
        return d.cfg

    keep = ops.DMA_WIDE_WAVES
    try:
        assert chosen(128, 64, 64) == 14 and chosen(64, 384, 128) == 14 and chosen(32, 256, 256) == 14   # >= 512 tiles x channel blocks
        assert chosen(32, 64, 64) == 11 and chosen(16, 512, 512) == 11                                     # small grids: cfg 11 (+ split-K)
        assert chosen(128, 64, 64, pre=True) == 11                                                         # fused prologue
        assert chosen(128, 64, 64, stride=2) == 15
        d32 = desc(128, 64, 64)
        d32.dtype = 0
        ops._choose_conv_cfg(d32, 128 ** 3, only=ops.DMA_CFGS)
        assert d32.cfg == 14                                                                               # fp32 too
        for gone in (21, 22):  # (round 6) the two 32x32x16 tile configurations left the library after the energy-metered no-go (experiments/conv_mw, conv_w8)
            with pytest.raises(ValueError):
                ops._choose_conv_cfg(desc(128, 64, 64), 128 ** 3, force_cfg=gone, only=ops.DMA_CFGS)
        ops.DMA_WIDE_WAVES = False
        assert chosen(128, 64, 64) == 11
    finally:
        ops.DMA_WIDE_WAVES = keep


def test_native_planners_accept_and_reject_geometries_without_a_gpu():
    """Host-side planning of the C-ABI library runs without a GPU: tile-configuration eligibility (gm_conv_lds_bytes: > 0 = bytes of LDS,
    -1 = configuration does not cover the geometry) for the LDS-DMA kernels incl. the sub-pixel up-sampling variant, and the weight-gradient
    planner (gm_conv_wgrad_workspace_bytes: -1 = not covered)."""
    import ctypes as C

    from generativemodels_amd import _native as nat

    lib = nat.lib()

    def conv_desc(**kw):
        d = nat.GmConvDesc()
        base = dict(N=1, Cin=64, Cout=64, Ds=16, Hs=16, Ws=16, Do=16, Ho=16, Wo=16, kd=3, kh=3, kw=3, sd=1, sh=1, sw=1, pd=1, ph=1, pw=1,
                    dd=1, dh=1, dw=1, in_mode=0, fd=1, fh=1, fw=1, dtype=1, ltd=2, lth=2, ltw=4, cfg=11, x_ld=64, y_ld=64)
        base.update(kw)
        for k, v in base.items():
            setattr(d, k, v)
        d.x, d.y, d.w = 0x1000, 0x2000, 0x3000  # aligned dummies: the planner only inspects them
        return d

    def lds(**kw):
        return lib.gm_conv_lds_bytes(C.byref(conv_desc(**kw)))

    assert lds() == 6 * 112 * 64 + 3 * 192 * 64 + 512                     # cfg 11: 3x3x3 stride 1
    assert lds(Cin=48) == -1                                          # C_in must be a multiple of the 32-channel K step (bf16)
    assert lds(kd=1, Ds=1, Do=1) == -1                                # 2-D convolutions stay on the other kernels
    assert lds(cfg=15, sd=2, sh=2, sw=2, Do=8, Ho=8, Wo=8, ltd=1) > 0 # stride-2 variant
    assert lds(cfg=15) == -1                                          # ... which does not take stride 1
    sub = dict(cfg=17, in_mode=3, kd=2, kh=2, kw=2, Do=32, Ho=32, Wo=32, pd=0, ph=0, pw=0)
    assert lds(**sub) == 5 * 96 * 64 + 36 * 1024 + 512 + 256 * 9 * 4  # sub-pixel up-sampling (2x2x2 kernels): four 8 KiB panels padded to the 36 KiB transpose scratch, + the placement table (round 5)
    assert lds(**dict(sub, Do=31)) == -1
    assert lds(**dict(sub, in_mode=1)) == -1
    assert lds(in_mode=3, kd=2, kh=2, kw=2, Do=32, Ho=32, Wo=32) == -1  # in_mode 3 exists for configuration 17 only
    assert lds(cfg=21) == -1 and lds(cfg=22) == -1                    # (round 6) no longer in the library

    def wgrad_bytes(**kw):
        d = nat.GmWgradDesc()
        base = dict(N=1, Cin=64, Cout=64, Ds=16, Hs=16, Ws=16, Do=16, Ho=16, Wo=16, kd=3, kh=3, kw=3, stride=1, pd=1, ph=1, pw=1, dtype=1,
                    accumulate=0, x_ld=64, gy_ld=64)
        base.update(kw)
        for k, v in base.items():
            setattr(d, k, v)
        d.x, d.gy = 0x1000, 0x2000
        return lib.gm_conv_wgrad_workspace_bytes(C.byref(d))

    tiles = (16 // 2) * (16 // 4) * 1                                  # 2 x 4 x 32 voxel tiles
    assert wgrad_bytes() == min(256 // 3, tiles) * 3 * 9 * 64 * 64 * 4 # one work-group per CU, capped by the tile count
    assert wgrad_bytes(kd=1, kh=1, kw=1, pd=0, ph=0, pw=0) > 0         # 1x1 convolutions / nn.Linear: flat mode
    assert wgrad_bytes(stride=2, Do=8, Ho=8, Wo=8) > 0
    assert wgrad_bytes(Cin=60) == -1                                   # ragged channel counts are padded by ops.conv_wgrad, not here
    assert wgrad_bytes(kd=3, kh=5, kw=5) == -1
    assert wgrad_bytes(stride=3) == -1

    # split-K: the combine kernel's row blocks follow the volume -- as few iterations as keep the launch (= the statistic partials of the output)
    # at <= 256 blocks per sample, at most 8 (round 4; it was 8 iterations whatever the size: an 8^3 x 256-channel output ran on 8 blocks)
    def splitk_slots(cout, sp, ks, dtype=1):
        d = conv_desc(Cin=256, Cout=cout, Ds=sp, Hs=sp, Ws=sp, Do=sp, Ho=sp, Wo=sp, y_ld=cout, dtype=dtype, ksplit=ks)
        d.kpartial = 0x8000
        assert lib.gm_conv_splitk_workspace_bytes(C.byref(d)) == ks * sp ** 3 * cout * 4
        return lib.gm_conv_stats_slots(C.byref(d))

    assert splitk_slots(256, 8, 8) == 512 // 8            # 8 rows in flight x 1 iteration: 64 blocks (was 8)
    assert splitk_slots(128, 16, 4) == 256                # 16 rows x 1 iteration
    assert splitk_slots(64, 32, 2) == 256                 # 32 rows x 4 iterations
    assert splitk_slots(64, 64, 2) == 64 ** 3 // (32 * 8) # large volumes: the 8-iteration cap, the table is compacted afterwards
    assert splitk_slots(64, 32, 2, dtype=0) == 256        # fp32: 16 rows x 8 iterations

    # EMA codebook statistics: the workspace is one [K x (D + 1)] table per token range of 1024 (fewer, longer ranges beyond 1024 of them)
    assert lib.gm_vq_ema_stats_workspace_elems(0, 16, 8) == 16 * 9
    assert lib.gm_vq_ema_stats_workspace_elems(5000, 256, 32) == 5 * 256 * 33
    assert lib.gm_vq_ema_stats_workspace_elems(1 << 21, 256, 32) == 1024 * 256 * 33

    # attention: the output statistics ride on the split-KV merge kernel (one head, LDS-DMA geometry), the V image may come packed
    def attn_desc(**kw):
        d = nat.GmAttnDesc()
        base = dict(B=1, H=1, Lq=512, Lk=512, dh=256, scale=1.0, dtype=1, q_ld=768, k_ld=768, v_ld=768, o_ld=256, causal=0)
        base.update(kw)
        for k, v in base.items():
            setattr(d, k, v)
        d.q, d.k, d.v, d.o = 0x1000, 0x1200, 0x1400, 0x9000
        return d

    assert lib.gm_attention_workspace_bytes(C.byref(attn_desc())) > 256 * 512 * 2          # V^T image + the split-KV partial states
    assert lib.gm_attention_stats_slots(C.byref(attn_desc())) == 512 * (256 // 8) // 256    # one merge block per 256 (query, 8-channel) items
    assert lib.gm_attention_stats_slots(C.byref(attn_desc(Lq=4096, Lk=4096, dh=128))) == 256  # capped: the consumer reads the table without a compaction
    assert lib.gm_attention_stats_slots(C.byref(attn_desc(H=2, dh=128))) == 0                # several heads: the statistics pass stays
    assert lib.gm_attention_stats_slots(C.byref(attn_desc(Lq=32768, Lk=32768))) == 0         # enough query tiles for the chip: no key slices, no merge kernel
    assert lib.gm_attention_stats_slots(C.byref(attn_desc(dh=80, q_ld=240, k_ld=240, v_ld=240, o_ld=80))) == 0  # not an LDS-DMA head dim


def test_training_api_has_no_cpu_fallback_and_checks_its_arguments():
    """generativemodels_amd.autograd / forward_train on CPU tensors raise (no eager fallback); argument checks run before any kernel."""
    from generativemodels_amd import autograd as A
    from generativemodels_amd import ops

    x = torch.zeros(1, 4, 4, 4, 8, requires_grad=True)
    w = torch.zeros(8, 8, 3, 3, 3, requires_grad=True)
    with pytest.raises(RuntimeError, match="MI355X"):
        A.conv(x, w, None, kernel=3, padding=1)
    with pytest.raises(RuntimeError, match="MI355X"):
        A.group_norm_act(x, None, None, 4, 1e-6, "silu")
    with pytest.raises(RuntimeError, match="MI355X"):
        A.attention(torch.zeros(1, 8, 16), torch.zeros(1, 8, 16), torch.zeros(1, 8, 16), 2, 0.35)
    with pytest.raises(ValueError):
        A.linear(torch.zeros(4, 8), torch.zeros(8, 8))             # (N, L, C) expected
    with pytest.raises((RuntimeError, ValueError)):
        ops.conv_wgrad(torch.zeros(1, 4, 4, 8), torch.zeros(1, 4, 4, 4, 8), 3, 1, 1)  # rank mismatch / CPU tensors
    unet = DiffusionModelUNet(2, 1, 1, num_channels=(8, 16), attention_levels=(False, True), num_res_blocks=1, norm_num_groups=8, num_head_channels=8)
    assert unet.supports_training()
    with pytest.raises(RuntimeError, match="MI355X"):
        unet.forward_train(torch.zeros(1, 1, 8, 8), torch.tensor([3]))
    with pytest.raises(ValueError):
        unet.forward_train(torch.zeros(1, 1, 8, 8), torch.tensor([[3]]))
    with pytest.raises(ValueError):
        unet.forward_train(torch.zeros(1, 1, 8, 8), torch.tensor([3]), context=torch.zeros(1, 2, 4))   # no conditioning in this network


def test_autocast_region_selects_the_compute_dtype_and_wants_grad_follows_the_input():
    """Host logic of the mixed-precision switch (ops.autocast: fp32 master parameters, bf16 compute -- the reference's autocast training,
    ddpm_training_ddp.py:129,253-270) and of the train / inference dispatch (`_blocks.wants_grad`)."""
    import threading
    import warnings

    import generativemodels_amd as gm
    from generativemodels_amd import ops
    from generativemodels_amd.networks.nets import _blocks

    assert ops.autocast_dtype() is None and ops.compute_dtype(torch.float32) == torch.float32
    with gm.autocast(torch.bfloat16):
        assert ops.autocast_dtype() == torch.bfloat16 and ops.compute_dtype(torch.float32) == torch.bfloat16
        with gm.autocast(enabled=False):  # an inner region switches it off
            assert ops.autocast_dtype() is None
        seen = []
        th = threading.Thread(target=lambda: seen.append(ops.autocast_dtype()))  # thread-local, like torch.autocast
        th.start(); th.join()
        assert seen == [None]
        assert ops.autocast_dtype() == torch.bfloat16
    assert ops.autocast_dtype() is None
    with pytest.raises(TypeError):
        gm.autocast(torch.float16)  # fp32 / bf16 kernels only
    # outside a region a dtype mismatch at a network's entry is an error, as before
    with pytest.raises(TypeError):
        ops.entry_cast(torch.zeros(2, dtype=torch.bfloat16), torch.float32)
    assert ops.entry_cast(torch.zeros(2), torch.float32).dtype == torch.float32

    m = DiffusionModelUNet(2, 1, 1, num_channels=(8,), attention_levels=(False,), num_res_blocks=1, norm_num_groups=8)
    x = torch.zeros(1, 1, 8, 8)
    xg = torch.zeros(1, 1, 8, 8, requires_grad=True)
    assert _blocks.wants_grad(m.train(), x) and _blocks.wants_grad(m.train(), xg)
    with torch.no_grad():
        assert not _blocks.wants_grad(m, xg)
    for p in m.parameters():
        p.requires_grad_(False)
    # frozen network: the gradient still has to reach an input that asks for it (a pixel loss through a frozen decoder, ControlNet training)
    assert _blocks.wants_grad(m, xg) and _blocks.wants_grad(m.eval(), xg) and not _blocks.wants_grad(m.train(), x)
    for p in m.parameters():
        p.requires_grad_(True)
    _blocks._warned_eval_grad = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert not _blocks.wants_grad(m.eval(), x)   # eval + trainable parameters + gradients enabled: inference path, said out loud once
        assert not _blocks.wants_grad(m.eval(), x)
    assert len([i for i in w if "without grad_fn" in str(i.message)]) == 1


def test_stride2_phase_tap_tables_reproduce_a_k4_stride2_weight_gradient():
    """ops.stride2_phase_taps (the weight gradient of a kernel-4 / stride-2 convolution assembled from 3-tap stride-1 weight gradients over the
    phase images of its input: VQ-VAE down- / up-sampling, vqvae.py:127-150,244-261) against a brute-force 1-D weight gradient, every
    padding the decomposition admits, even and odd extents."""
    import numpy as np

    from generativemodels_amd import ops

    rng = np.random.default_rng(1)
    for pad in (0, 1, 2):
        for X in (10, 11):
            O = (X + 2 * pad - 4) // 2 + 1
            x, gy = rng.standard_normal(X), rng.standard_normal(O)
            want = np.zeros(4)
            for o in range(O):
                for k in range(4):
                    i = 2 * o + k - pad
                    if 0 <= i < X:
                        want[k] += gy[o] * x[i]
            got = np.zeros(4)
            taps = ops.stride2_phase_taps(pad)
            for r in (0, 1):
                rho, t0, t1 = taps[r]
                ph = x[rho::2]
                g3 = np.zeros(3)   # a 3-tap, padding-1, stride-1 weight gradient over the phase image
                for o in range(O):
                    for t in range(3):
                        i = o + t - 1
                        if 0 <= i < len(ph):
                            g3[t] += gy[o] * ph[i]
                got[r], got[r + 2] = g3[t0], g3[t1]
            assert np.allclose(got, want), (pad, X)


def test_trainable_parameter_derivatives_are_remade_inside_a_training_capture():
    """ops._cached serves packed panels / fp32 copies per (parameter, version); while a training step is being captured into a HIP graph
    (graphs.GraphedForwardBackward enters ops.refresh_trainable_derivatives) the derivatives of TRAINABLE parameters must be re-made inside the
    capture -- a cached panel would be baked into the graph and the replays would keep reading the weights of the capture step -- while frozen
    parameters (inference graphs, a frozen autoencoder) keep the cache.  Pure host logic."""
    import torch
    from generativemodels_amd import ops
    calls = []

    def make():
        calls.append(1)
        return object()

    trainable = torch.nn.Parameter(torch.zeros(4))
    frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
    a = ops._cached(trainable, "t", make)
    assert ops._cached(trainable, "t", make) is a and len(calls) == 1
    with ops.refresh_trainable_derivatives():
        b = ops._cached(trainable, "t", make)
        c = ops._cached(trainable, "t", make)
        assert b is not a and c is not b and len(calls) == 3
        f1 = ops._cached(frozen, "t", make)
        assert ops._cached(frozen, "t", make) is f1 and len(calls) == 4
        with ops.refresh_trainable_derivatives():  # nests
            pass
        assert ops._REFRESH_TRAINABLE[0]
    assert not ops._REFRESH_TRAINABLE[0]
    assert ops._cached(trainable, "t", make) is a  # the eager cache entry survived the capture
    with torch.no_grad():
        trainable.add_(1.0)  # an optimizer step: the version changes, the entry is stale
    assert ops._cached(trainable, "t", make) is not a


def test_graphed_forward_backward_rejects_what_it_cannot_capture():
    import pytest
    import torch
    import generativemodels_amd as gm
    lin = torch.nn.Linear(4, 4)
    with pytest.raises(ValueError):  # host tensors cannot be replayed from device memory
        gm.GraphedForwardBackward(lambda x: lin(x).sum(), (torch.zeros(2, 4),), lin.parameters())
    for p in lin.parameters():
        p.requires_grad_(False)
    with pytest.raises(ValueError):  # nothing to differentiate
        gm.GraphedForwardBackward(lambda x: lin(x).sum(), (torch.zeros(2, 4),), lin.parameters())


def test_bench_module_imports_and_its_power_sampler_degrades_without_a_gpu():
    """bench.py must import on a CPU-only host (tests import its C2 / rerandomize_zero_params) and its power sampler must come back with None --
    not raise -- where neither librocm_smi64 nor the amdgpu hwmon node reports a GPU."""
    import bench

    assert bench.C2["num_channels"] == (64, 128, 256) and len(bench.kernel_source_sha()) == 16
    ps = bench.PowerSampler(0)
    ps.start()
    out = ps.stop()
    assert out is None or (out["mean_w"] > 0 and out["samples"] >= 1)
    assert set(bench.MFMA_BF16_SUSTAINED_TFLOPS.values()) == {1863.0, 1490.0, 1621.0}  # profiles/r04_mfma_power_ceiling.txt


def test_bf16_noise_table_reproduces_torch_randn_from_the_generator_bytes():
    """host_noise.py: torch's CPU bf16 `randn` (what the reference's DDPMScheduler.step draws for a bf16 chain, ddpm.py:244-248) is a function of byte pairs of the
    generator's draws inside blocks of 16.  The table (built with torch's own bf16 operators) and the host-side lookup -- the checker of the device kernel --
    against torch.randn itself: several seeds and shapes, the default generator, signed zeros (u1 = 1), and the generator must end in the same state.  Without a GPU
    (or for other dtypes / ragged sizes) `host_noise.randn` IS torch.randn."""
    from generativemodels_amd import host_noise as H
    assert H.table_matches_torch()
    assert H.bf16_normal_table().shape == (256, 256) and H.bf16_normal_table().dtype == torch.int32
    for seed in (0, 1, 77):
        for shape in ((16, 1, 64, 64), (2, 3, 8), (1, 16), (5, 4, 32, 32)):
            g1, g2 = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed)
            want = torch.randn(shape, dtype=torch.bfloat16, generator=g1)
            bits = torch.empty(want.numel(), dtype=torch.uint8).random_(generator=g2)
            got = H.normal_bf16_from_bits_host(bits).reshape(shape)
            assert torch.equal(want.view(torch.int16), got.view(torch.int16)) and torch.equal(g1.get_state(), g2.get_state()), (seed, shape)
    torch.manual_seed(9)
    want = torch.randn(64, dtype=torch.bfloat16)
    torch.manual_seed(9)
    assert torch.equal(want.view(torch.int16), H.normal_bf16_from_bits_host(torch.empty(64, dtype=torch.uint8).random_()).view(torch.int16))
    zero_pairs = torch.zeros(16, dtype=torch.uint8)  # u1 = 1: r = sqrt(-0); `* std + mean` makes every result +0
    assert torch.equal(H.normal_bf16_from_bits_host(zero_pairs).view(torch.int16), torch.zeros(16, dtype=torch.int16))
    with pytest.raises(ValueError):
        H.normal_bf16_from_bits_host(torch.zeros(24, dtype=torch.uint8))
    for dtype, shape in ((torch.bfloat16, (4, 16)), (torch.bfloat16, (3, 5)), (torch.float32, (4, 16))):  # no device: the plain draw, same stream
        g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
        assert torch.equal(H.randn(shape, dtype, g1, "cpu"), torch.randn(shape, dtype=dtype, generator=g2)) and torch.equal(g1.get_state(), g2.get_state())


def test_bench_self_launch_argv_and_environment():
    """`python bench.py --gpus N` as the driver calls it (no launcher environment): N > 1 -- or GM_BENCH_SELF_LAUNCH=1 at N = 1 -- re-executes
    itself under torch.distributed.run with one rank per GPU, rendezvous on 127.0.0.1, the original arguments handed on unchanged; a process
    that already IS a rank (WORLD_SIZE / RANK set by the launcher) never launches again."""
    import sys

    import bench

    assert not bench.needs_self_launch(1, {})
    assert bench.needs_self_launch(2, {}) and bench.needs_self_launch(8, {"PATH": "/usr/bin"})
    assert bench.needs_self_launch(1, {"GM_BENCH_SELF_LAUNCH": "1"})
    assert not bench.needs_self_launch(8, {"WORLD_SIZE": "8", "RANK": "3"})
    assert not bench.needs_self_launch(1, {"GM_BENCH_SELF_LAUNCH": "1", "RANK": "0", "WORLD_SIZE": "1"})  # the child of a forced self-launch
    argv = ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    cmd, env = bench.self_launch_command(8, argv, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(bench.__file__ if bench.__file__ in cmd else __import__("os").path.abspath(bench.__file__))
    assert cmd[i + 1:] == argv
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["GM_BENCH_SELF_LAUNCHED"] == "1" and int(env["OMP_NUM_THREADS"]) >= 1
    cmd2, _ = bench.self_launch_command(2, [])  # a free port is picked when none is given
    assert 1024 < int(cmd2[cmd2.index("--master-port") + 1]) < 65536


def test_attention_backward_policy_and_timestep_row_hand_over_host_side():
    """Round 5 host logic without a GPU: which attention shapes train through the fused LDS-DMA backward (`autograd._fused_backward_serves`: bf16, head
    dim 64 / 128 / 256, at least ATTENTION_BWD_FUSED_MIN_TOKENS on either side), that the fused entry point has no CPU fallback, and the split of a
    stacked timestep row (`DiffusionModelUNet.time_rows_table` / `_time_rows_row`) into per-ResnetBlock views -- rejecting a row of the wrong width,
    None for class-conditional networks."""
    from generativemodels_amd import autograd as A, ops
    bf, f32 = torch.bfloat16, torch.float32
    q = lambda l, c, dt=bf: torch.zeros((1, l, c), dtype=dt)
    assert A._fused_backward_serves(q(4096, 256), q(4096, 256), 1)
    assert A._fused_backward_serves(q(4096, 512), q(77, 512), 8)                       # cross-attention: the longer side counts
    assert A._fused_backward_serves(q(256, 128), q(256, 128), 2)
    assert not A._fused_backward_serves(q(255, 128), q(128, 128), 2)                   # below the measured bound
    assert not A._fused_backward_serves(q(4096, 256, f32), q(4096, 256, f32), 1)       # fp32: the fp32-MFMA kernels
    assert not A._fused_backward_serves(q(4096, 256), q(4096, 256), 8)                 # head dim 32: the composed bf16 path
    assert not A._fused_backward_serves(q(4096, 192), q(4096, 192), 1)                 # head dim 192: padded to 256 by the caller first
    with pytest.raises(RuntimeError, match="MI355X"):
        ops.attention_backward_fused(q(512, 64), q(512, 64), q(512, 64), q(512, 64), q(512, 64), 1, 0.125)  # CPU tensors: no fallback
    torch.manual_seed(0)
    m = DiffusionModelUNet(spatial_dims=2, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(32, 64), attention_levels=(False, False),
                           norm_num_groups=32)
    blocks = m._resnets_in_order()
    total = sum(b.out_channels for b in blocks)
    row = torch.arange(total, dtype=f32).unsqueeze(0)
    parts = m._temb_split(row)
    off = 0
    for b in blocks:
        assert torch.equal(parts[id(b)], row[:, off:off + b.out_channels]) and parts[id(b)].data_ptr() == row[:, off:].data_ptr()
        off += b.out_channels
    with pytest.raises(ValueError):
        m._temb_split(row[:, :-1])
    mc = DiffusionModelUNet(spatial_dims=2, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(32, 32), attention_levels=(False, False),
                            norm_num_groups=32, num_class_embeds=3)
    assert mc.time_rows_table(torch.zeros(4)) is None
    assert DiffusionInferer.BATCHED_TIME_ROWS is True
