"""GPU (-m gpu): parity against the CPU oracle AT BASELINE.json's REAL SIZES (VERDICT r1 "weak #1": until now the full-size path was only
compared with itself).  The oracle (oracle/restatement.py, pinned to the unmodified reference by tests/test_oracle_golden.py) runs whole
tensors of the headline size on the box's host cores -- about a minute per UNet forward -- so these are the slow tests of the suite:

  * C2 (configs[1]): DiffusionModelUNet(64,128,256) on 1x1x128^3 at t in {980, 500, 20}: GPU fp32 forward within the fp32 bar
    1e-4 * max(1, |ref|_inf), GPU bf16 within the bf16 bar of SURVEY.md 8(c)(3), and the teacher-forced DDIM step x_t -> x_{t-1}
    (both sides fed the same x_t; reference: inferers/inferer.py:119-137, schedulers/ddim.py:156-237);
  * C3 (configs[2]): the brain-bundle AutoencoderKL (64,128,128,128; latent 4) encode of a 1x1x128^3 volume and decode of a 1x4x16^3
    latent, whole tensors (nets/autoencoderkl.py:718-799);
  * C5 (configs[4]): VQVAE encode of a 128^3 volume (nets/vqvae.py:414-437) + DecoderOnlyTransformer(257, 4096, 256, 12, 8) at its real
    width: 64 tokens of the greedy sampling trajectory, teacher-forced through the KV-cache step (inferers/inferer.py:1183-1245).
The oracle's thread count is capped (ORACLE_THREADS): oneDNN oversubscribes on the 256-core box (71 s per forward at 256 threads)."""
import os

import pytest
import torch

import restatement as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
ORACLE_THREADS = min(64, os.cpu_count() or 1)


def _oracle(fn):
    keep = torch.get_num_threads()
    torch.set_num_threads(ORACLE_THREADS)
    try:
        with torch.no_grad():
            return fn()
    finally:
        torch.set_num_threads(keep)


def _fp32_bar(got, want, what):
    got, want = got.detach().float().cpu(), want.float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    print(f"[parity] {what}: max|err| {err:.3e} (bar {1e-4 * scale:.3e}, |ref|_inf {scale:.3g})")
    assert err <= 1e-4 * scale, f"{what}: max|err| {err:.3e} > {1e-4 * scale:.3e}"


def _bf16_bar(got, want, what):
    got, want = got.detach().float().cpu(), want.float()
    sigma = max(want.std().item(), 1e-3)
    err = (got - want).abs()
    print(f"[parity] {what}: mean|err| {err.mean().item():.3e} max|err| {err.max().item():.3e} (sigma {sigma:.3g})")
    assert err.mean().item() <= 2e-2 * sigma and err.max().item() <= 0.2 * sigma, \
        f"{what}: mean|err| {err.mean().item():.3e}, max|err| {err.max().item():.3e}, sigma {sigma:.3e}"


# ---- C2 ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2():
    from bench import C2, rerandomize_zero_params
    from generativemodels_amd.networks.nets import DiffusionModelUNet

    torch.manual_seed(0)
    m = DiffusionModelUNet(**C2).eval()
    sd = rerandomize_zero_params({k: v.clone() for k, v in m.state_dict().items()})
    x = torch.randn((1, 1, 128, 128, 128), generator=torch.Generator().manual_seed(7))

    def build(dtype):
        net = DiffusionModelUNet(**C2).eval()
        net.load_state_dict(sd)
        return net.to(DEV, dtype)

    return dict(cfg=C2, sd=sd, x=x, m32=build(torch.float32), m16=build(torch.bfloat16))


@pytest.mark.parametrize("t", [980, 500, 20])
def test_c2_forward_and_teacher_forced_ddim_step_match_the_oracle_at_full_size(c2, t):
    from generativemodels_amd.networks.schedulers import DDIMScheduler

    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(50)
    assert t in [int(v) for v in sched.timesteps]
    x, sd, cfg = c2["x"], c2["sd"], c2["cfg"]
    eps_ref = _oracle(lambda: R.unet_forward(sd, cfg, x, torch.tensor([float(t)])))
    prev_ref, _ = R.ddim_step(sched.alphas_cumprod, 1000, 50, eps_ref, t, x, clip_sample=False)
    ts = torch.tensor([float(t)], device=DEV)
    # fp32 storage, exact-fp32 MFMA: the fp32 bar
    eps32 = c2["m32"](x.to(DEV), ts)
    _fp32_bar(eps32, eps_ref, f"C2 1x1x128^3 fp32 forward t={t}")
    prev32, _ = sched.step(eps32, t, x.to(DEV))
    _fp32_bar(prev32, prev_ref, f"C2 teacher-forced DDIM step t={t} (fp32)")
    # bf16 storage (the benchmarked path): the bf16 bar against the SAME fp32 oracle output
    xb = x.to(DEV, torch.bfloat16)
    eps16 = c2["m16"](xb, ts)
    _bf16_bar(eps16, eps_ref, f"C2 1x1x128^3 bf16 forward t={t}")
    prev16, _ = sched.step(eps16, t, xb)
    _bf16_bar(prev16, prev_ref, f"C2 teacher-forced DDIM step t={t} (bf16)")


def test_c2_full_size_forward_against_outputs_of_the_reference_itself(c2):
    """tests/golden/c2_fullsize_ref.pt (oracle/make_golden_c2_fullsize.py): the UNMODIFIED reference's own C2 predictions at 1x1x128^3, t = 980 / 500 / 20,
    kept as the every-4th-voxel lattice + whole-tensor mean / std / max-abs / axis projections -- no restatement between the HIP path and the
    reference at the headline size -- and the reference's OWN bf16 error at this size: SURVEY 8(c)(3)'s third clause
    err(ours_bf16) <= 1.5 err(ref_bf16) at C2 size (weak 1(b)).  Weights / noise are rebuilt from the seeds the fixture was made with."""
    from make_golden_c2_fullsize import sd_checksum, whole_tensor_summary

    from _util import GOLDEN

    fx = torch.load(os.path.join(GOLDEN, "c2_fullsize_ref.pt"), weights_only=False)
    assert abs(sd_checksum(c2["sd"]) - fx["sd_checksum"]) <= 1e-9 * fx["sd_checksum"], "the rebuilt state_dict is not the one the fixture was generated with"
    L = fx["lattice"]
    x = c2["x"]
    for t, ref in fx["fp32"].items():
        eps32 = c2["m32"](x.to(DEV), torch.tensor([float(t)], device=DEV)).float().cpu()
        _fp32_bar(eps32[..., ::L, ::L, ::L], ref["lattice"], f"C2 1x1x128^3 fp32 forward t={t} vs the reference's own output (lattice of {ref['lattice'].numel()} voxels)")
        got = whole_tensor_summary(eps32)
        n_line = 128 * 128  # voxels behind one entry of a projection: the fp32 bar per voxel bounds the sum
        for k in ("proj_d", "proj_h", "proj_w"):
            err = (got[k].double() - ref[k].double()).abs().max().item()
            assert err <= 1e-4 * max(1.0, ref["absmax"]) * n_line, (t, k, err)
        assert abs(got["mean"] - ref["mean"]) <= 1e-4 and abs(got["std"] - ref["std"]) <= 1e-4 * ref["std"] and abs(got["absmax"] - ref["absmax"]) <= 1e-4 * ref["absmax"], (t, got["mean"], ref["mean"])
    ref16 = fx["bf16"][500]
    eps16 = c2["m16"](x.to(DEV, torch.bfloat16), torch.tensor([500.0], device=DEV)).float().cpu()
    err = (eps16[..., ::L, ::L, ::L] - fx["fp32"][500]["lattice"]).abs()
    print(f"[parity] C2 1x1x128^3 bf16 t=500 vs the reference's fp32 output: ours mean|err| {err.mean().item():.4e} max {err.max().item():.4e}; "
          f"the reference's own bf16 run: mean {ref16['lattice_mean_err']:.4e} max {ref16['lattice_max_err']:.4e} (sigma {ref16['sigma']:.3f})")
    assert err.mean().item() <= 1.5 * ref16["lattice_mean_err"], f"ours bf16 mean|err| {err.mean().item():.3e} vs reference bf16 {ref16['lattice_mean_err']:.3e}"
    assert err.max().item() <= 2.0 * ref16["lattice_max_err"], f"ours bf16 max|err| {err.max().item():.3e} vs reference bf16 {ref16['lattice_max_err']:.3e}"


# ---- C3 ------------------------------------------------------------------------------------------------------------------------------
AEKL_BRAIN = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, num_channels=(64, 128, 128, 128), num_res_blocks=2,
                  attention_levels=(False, False, False, False), with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)


def test_c3_autoencoderkl_encode_decode_match_the_oracle_at_128_cubed():
    from generativemodels_amd.networks.nets import AutoencoderKL

    torch.manual_seed(0)
    m = AutoencoderKL(**AEKL_BRAIN).eval()
    sd = R.synthetic_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=11)
    m.load_state_dict(sd)
    x = torch.randn((1, 1, 128, 128, 128), generator=torch.Generator().manual_seed(9))
    z = torch.randn((1, 4, 16, 16, 16), generator=torch.Generator().manual_seed(8))
    mu_ref, sigma_ref = _oracle(lambda: R.aekl_encode(sd, AEKL_BRAIN, x))
    rec_ref = _oracle(lambda: R.aekl_decode(sd, AEKL_BRAIN, z))
    m32 = m.to(DEV)
    mu, sigma = m32.encode(x.to(DEV))
    _fp32_bar(mu, mu_ref, "C3 AutoencoderKL encode z_mu (1x1x128^3, fp32)")
    _fp32_bar(sigma, sigma_ref, "C3 AutoencoderKL encode z_sigma (fp32)")
    _fp32_bar(m32.decode(z.to(DEV)), rec_ref, "C3 AutoencoderKL decode to 1x1x128^3 (fp32)")
    mb = AutoencoderKL(**AEKL_BRAIN).eval()
    mb.load_state_dict(sd)
    mb = mb.to(DEV, torch.bfloat16)
    _bf16_bar(mb.decode(z.to(DEV, torch.bfloat16)), rec_ref, "C3 AutoencoderKL decode (bf16)")
    mu16, _ = mb.encode(x.to(DEV, torch.bfloat16))
    _bf16_bar(mu16, mu_ref, "C3 AutoencoderKL encode z_mu (bf16)")


# ---- C5 ------------------------------------------------------------------------------------------------------------------------------
def test_c5_vqvae_encode_and_transformer_trajectory_match_the_oracle_at_real_dims():
    from generativemodels_amd.networks.nets import VQVAE, DecoderOnlyTransformer

    vq_cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, num_embeddings=256, embedding_dim=32)
    tr_cfg = dict(num_tokens=257, max_seq_len=4096, attn_layers_dim=256, attn_layers_depth=12, attn_layers_heads=8)
    torch.manual_seed(0)
    vq = VQVAE(**vq_cfg).eval()
    vsd = R.synthetic_state_dict({k: tuple(v.shape) for k, v in vq.state_dict().items() if v.is_floating_point()}, seed=21)
    vsd = {**{k: v.clone() for k, v in vq.state_dict().items()}, **vsd}
    vq.load_state_dict(vsd)
    x = torch.randn((1, 1, 128, 128, 128), generator=torch.Generator().manual_seed(5))
    z_ref = _oracle(lambda: R.vqvae_encode(vsd, vq_cfg, x))
    idx_ref, _ = R.vq_index_quantize(vsd, z_ref)
    vqd = vq.to(DEV)
    z = vqd.encode(x.to(DEV))
    _fp32_bar(z, z_ref, "C5 VQVAE encode 1x1x128^3 -> 1x32x16^3 (fp32)")
    idx = vqd.index_quantize(x.to(DEV)).cpu()
    assert idx.shape == idx_ref.shape == (1, 16, 16, 16)
    diff = (idx != idx_ref).reshape(-1).nonzero().reshape(-1)
    if diff.numel():  # integer work is bit-exact except at exact / near ties of the fp32 distances (vector_quantizer.py:109-116)
        emb = vsd["quantizer.quantizer.embedding.weight"].double()
        flat = z_ref.double().permute(0, 2, 3, 4, 1).reshape(-1, emb.shape[1])[diff]
        d = (flat ** 2).sum(1, keepdim=True) + (emb ** 2).sum(1)[None] - 2 * flat @ emb.t()
        rows = torch.arange(diff.numel())
        gap = (d[rows, idx.reshape(-1)[diff]] - d[rows, idx_ref.reshape(-1)[diff]]).abs()
        assert diff.numel() <= 4 and gap.max().item() <= 1e-4 * d.abs().max().item(), (diff.numel(), gap.max().item())

    tr = DecoderOnlyTransformer(**tr_cfg).eval()
    tsd = R.synthetic_state_dict({k: tuple(v.shape) for k, v in tr.state_dict().items() if v.is_floating_point()}, seed=22)
    tsd = {**{k: v.clone() for k, v in tr.state_dict().items()}, **tsd}
    tr.load_state_dict(tsd)
    trd = tr.to(DEV)
    # the oracle's sampling trajectory (reference loop: one full forward of the growing prefix per token, last-position logits, BOS
    # masked); tokens are DRAWN (seeded CPU generator, temperature 8) so the prefix is diverse -- greedy decoding of random-init
    # weights repeats one token -- and the greedy choice is checked per step wherever the oracle's top-2 gap is not a near-tie
    bos, ntok = 256, 64
    gen = torch.Generator().manual_seed(33)
    seq = torch.full((1, 1), bos, dtype=torch.long)
    ref_logits, gaps, greedy = [], [], []
    for _ in range(ntok):
        lg = _oracle(lambda: R.transformer_forward(tsd, tr_cfg, seq))[:, -1, :]
        ref_logits.append(lg)
        probs = R.transformer_sample_probs(lg.clone(), 1.0, None, bos)
        top2 = torch.topk(probs, 2, dim=-1)
        gaps.append((top2.values[0, 0] - top2.values[0, 1]).item() / top2.values[0, 0].item())
        greedy.append(int(top2.indices[0, 0]))
        seq = torch.cat([seq, torch.multinomial(R.transformer_sample_probs(lg.clone(), 8.0, None, bos), 1, generator=gen)], dim=1)
    assert len(set(seq[0].tolist())) > 16  # a diverse prefix
    cache = trd.new_cache(1, DEV)
    worst = 0.0
    for t in range(ntok):  # teacher-forced along the oracle's tokens, through the KV cache
        lg = trd.step(seq[:, t:t + 1].to(DEV), t, cache, None).float().cpu()
        scale = max(1.0, ref_logits[t].abs().max().item())
        err = (lg - ref_logits[t]).abs().max().item()
        worst = max(worst, err / scale)
        assert err <= 1e-4 * scale, f"C5 transformer step {t}: max|err| {err:.3e}"
        m = lg.clone()
        m[:, bos] = -float("inf")
        if gaps[t] > 1e-3:
            assert int(m.argmax(-1)) == greedy[t], f"greedy token {t} differs from the oracle's (relative top-2 gap {gaps[t]:.3e})"
    print(f"[parity] C5 transformer (257, 4096, 256, 12, 8): {ntok} teacher-forced KV-cache steps, worst relative logit error {worst:.3e}")
    # bf16 (the benchmarked dtype): last-step logits within the bf16 bar
    trb = DecoderOnlyTransformer(**tr_cfg).eval()
    trb.load_state_dict(tsd)
    trb = trb.to(DEV, torch.bfloat16)
    full = trb(seq[:, :ntok].to(DEV)).float().cpu()
    _bf16_bar(full[0], torch.cat(ref_logits, 0), "C5 transformer full-prefix forward (bf16) vs the oracle's per-step logits")
