"""GPU (-m gpu): the parity spots round 3 left thin (VERDICT r3 "weak #1"), at BASELINE's real sizes, against the CPU oracle
(oracle/restatement.py, pinned to the unmodified reference by tests/test_oracle_golden.py):

  * `AutoencoderKL.encode` of a 1x1x256^3 volume, WHOLE tensor (z_mu and z_sigma, fp32 and bf16) -- the 29 ms that dominate every C4 step,
    incl. the asymmetric-pad stride-2 convolutions at byte offsets above 2^32 (nets/autoencoderkl.py:355-447, 731-767); round 3 compared
    level-0 spot voxels only;
  * `LatentDiffusionInferer.__call__` at C4's size: encode (1x1x256^3) -> sampling (the SAME device draw on both sides) -> add_noise ->
    latent UNet (1x4x32^3) (inferers/inferer.py:300-360, schedulers/scheduler.py:169-185);
  * a FREE-RUNNING 10-step DDIM chain of config C2 at 1x1x128^3 in fp32 (clip_sample off): ~4 minutes of oracle time on the
    host, so it only runs with GM_SLOW_TESTS=1 (measured values: profiles/r04_fullsize_parity_measured.txt).
Measured values are printed as `[parity] ...` lines."""
import os

import pytest
import torch

import restatement as R
from test_gpu_fullsize_oracle_r3 import AEKL_BRAIN, C3_SCHED, C3_UNET, _aekl_state, _bf16_bar, _build_aekl, _build_unet, _c3_unet_state, _fp32_bar, _oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def volume256():
    asd = _aekl_state()
    img = torch.randn((1, 1, 256, 256, 256), generator=torch.Generator().manual_seed(23)) * 0.5
    z_mu, z_sigma = _oracle(lambda: R.aekl_encode(asd, AEKL_BRAIN, img))
    return dict(asd=asd, img=img, z_mu=z_mu, z_sigma=z_sigma)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_c3_autoencoderkl_encode_whole_tensor_at_256_cubed(volume256, dtype):
    assert tuple(volume256["z_mu"].shape) == (1, 4, 32, 32, 32)
    ae = _build_aekl(volume256["asd"], dtype)
    with torch.no_grad():
        z_mu, z_sigma = ae.encode(volume256["img"].to(DEV, dtype))
    bar = _fp32_bar if dtype == torch.float32 else _bf16_bar
    name = "fp32" if dtype == torch.float32 else "bf16"
    bar(z_mu, volume256["z_mu"], f"C3/C4 AutoencoderKL.encode 1x1x256^3 -> z_mu 1x4x32^3, whole tensor ({name})")
    if dtype == torch.float32:
        bar(z_sigma, volume256["z_sigma"], f"C3/C4 AutoencoderKL.encode 1x1x256^3 -> z_sigma, whole tensor ({name})")
    else:  # sigma = exp(log_var / 2) turns a bf16-sized error of log_var into a RELATIVE error of sigma (up to 1.0 absolute where sigma is ~50): the
        # bf16 bar is applied to the quantity the network computes, the clamped log-variance
        bar(2.0 * torch.log(z_sigma.float()), 2.0 * torch.log(volume256["z_sigma"]), f"C3/C4 AutoencoderKL.encode 1x1x256^3 -> log-variance, whole tensor ({name})")
        rel = ((z_sigma.float().cpu() - volume256["z_sigma"]).abs() / volume256["z_sigma"]).max().item()
        print(f"[parity] C3/C4 AutoencoderKL.encode 1x1x256^3 -> z_sigma (bf16): max relative error {rel:.3e}")
        assert rel <= 0.1
    del ae
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_c4_latent_diffusion_inferer_call_at_real_size(volume256, dtype):
    """inferer(inputs=1x1x256^3, autoencoder, unet, noise=1x4x32^3, timesteps): the reparameterisation draw is made on the device by
    `randn_like` -- the same generator state is replayed for the oracle, so both sides add the same eps * sigma."""
    from generativemodels_amd.inferers import LatentDiffusionInferer
    from generativemodels_amd.networks.schedulers import DDPMScheduler

    usd = _c3_unet_state()
    sched = DDPMScheduler(**{k: v for k, v in C3_SCHED.items() if k != "clip_sample"})
    noise = torch.randn((1, 4, 32, 32, 32), generator=torch.Generator().manual_seed(29))
    t = torch.tensor([417])
    torch.manual_seed(1234)
    eps = torch.randn((1, 4, 32, 32, 32), device=DEV, dtype=dtype)  # what AutoencoderKL.sampling will draw under this seed

    def oracle():
        lat = volume256["z_mu"] + eps.float().cpu() * volume256["z_sigma"]
        noisy = R.add_noise(sched.alphas_cumprod, lat * 0.9, noise, t)
        return R.unet_forward(usd, C3_UNET, noisy, t.float())

    want = _oracle(oracle)
    ae, unet = _build_aekl(volume256["asd"], dtype), _build_unet(usd, dtype)
    inf = LatentDiffusionInferer(sched, scale_factor=0.9)
    torch.manual_seed(1234)
    with torch.no_grad():
        got = inf(inputs=volume256["img"].to(DEV, dtype), autoencoder_model=ae, diffusion_model=unet, noise=noise.to(DEV, dtype), timesteps=t.to(DEV))
    if dtype == torch.float32:
        _fp32_bar(got, want, "C4 LatentDiffusionInferer.__call__ 1x1x256^3 -> eps 1x4x32^3 (fp32)", factor=2.0)
    else:
        _bf16_bar(got, want, "C4 LatentDiffusionInferer.__call__ 1x1x256^3 -> eps 1x4x32^3 (bf16)", factor=1.5)
    del ae, unet
    torch.cuda.empty_cache()


@pytest.mark.parametrize("steps", [5, pytest.param(10, marks=pytest.mark.skipif(not os.environ.get("GM_SLOW_TESTS"), reason="~4 minutes of oracle time on the host: set GM_SLOW_TESTS=1"))])
def test_c2_free_running_ddim_chain_at_128_cubed_fp32(steps):
    """A FREE-RUNNING chain (no teacher forcing: every step consumes the device's own previous output) of the benchmark's model at the
    benchmark's size against the oracle's chain -- 5 steps run in the driver's suite (~100 s of oracle time), 10 under GM_SLOW_TESTS=1."""
    from bench import C2, rerandomize_zero_params
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.nets import DiffusionModelUNet
    from generativemodels_amd.networks.schedulers import DDIMScheduler

    torch.manual_seed(0)
    m = DiffusionModelUNet(**C2).eval()
    sd = rerandomize_zero_params({k: v.clone() for k, v in m.state_dict().items()})
    m.load_state_dict(sd)
    m = m.to(DEV)
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(steps)
    noise = torch.randn((1, 1, 128, 128, 128), generator=torch.Generator().manual_seed(7))
    want = _oracle(lambda: R.ddim_sample(sd, C2, noise, dict(alphas_cumprod=sched.alphas_cumprod, num_train_timesteps=1000, num_inference_steps=steps,
                                                              timesteps=sched.timesteps, clip_sample=False)))
    with torch.no_grad():
        got = DiffusionInferer(sched).sample(noise.to(DEV), m, sched, verbose=False)
    _fp32_bar(got, want, f"C2 free-running DDIM-{steps} chain 1x1x128^3 (fp32)", factor=5.0)
