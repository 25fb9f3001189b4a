"""bench.py -- headline benchmark of the MI355X sampling path (BASELINE.json: 3D volumes/sec, DDIM 50-step sample of
1x128^3 volumes; UNet forward ms/step), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N = 1 runs in this process; N > 1 re-executes itself under
                                                                   # torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config C2): DiffusionModelUNet(3, 1, 1, num_channels=(64,128,256), 2 res blocks, attention in the
mid block with one 256-wide head) in bf16, DDIMScheduler(1000, scaled_linear_beta 0.0005..0.0195, clip_sample=False),
set_timesteps(50), one 1x1x128^3 volume per GPU per step.  A "step" = one complete 50-step sampling chain (50 UNet forwards +
50 fused scheduler steps) with the noise already resident in HBM.  Random-init weights (all-zero parameters re-randomised so
the network is not the zero function), synthetic Gaussian noise input.  Volumes are independent chains: ranks share nothing
on the data path (weak scaling, no collective); the barrier / all-reduce below only brackets the timing.

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  "roofline":     the dominant kernel (implicit-GEMM convolution) -- algorithmic FLOPs per launch / HIP-event time per launch
                  over one instrumented forward, against the dense bf16 MFMA peak;
  "cpu_baseline": kind "port" = oracle/restatement.py (the reference algorithm restated on torch-CPU fp32, pinned to outputs of the unmodified
                  reference by tests/golden/ -- at this size by c2_fullsize_ref.pt; the reference itself is Python and does not travel to the
                  GPU box), timed on this box's host cores on a bounded sample (median of three of the 50 steps at full size, at the best
                  thread count of a sweep), extrapolated to volumes/s -- and its t = 500 prediction compared with the GPU forward of the
                  same volume and with the reference's own output of it."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
# What the chip SUSTAINS on bf16 MFMA at its 1400 W package power cap with random (non-zero) operands -- tools/mfma_power.hip on MI355X,
# profiles/r04_mfma_power_ceiling.txt: the clock is set by the power budget, and this benchmark runs AT the cap (profiles/r04_power_trace.txt).
# Reported next to the nominal peak; `roofline.frac` stays achieved / nominal peak.
MFMA_BF16_SUSTAINED_TFLOPS = {"operands in registers, no memory traffic": 1863.0, "one ds_read_b128 per 32x32x16 MFMA (this kernel's tap loop)": 1490.0,
                              "0.75 ds_read_b128 per MFMA, 2 work-groups per CU (the best LDS-fed loop measured)": 1621.0}


class PowerSampler:
    """Socket package power of the GPU this rank runs on, sampled every 50 ms on a host thread while the timed region runs: the same quantity
    `rocm-smi --showpower` prints (rsmi_dev_current_socket_power_get through librocm_smi64 -- a library call: no process is spawned, nothing is
    enqueued on the GPU); falls back to the amdgpu hwmon node (a slower moving average) and to None when neither is there."""

    def __init__(self, local_rank: int = 0):
        import ctypes
        import glob
        self.kind, self.cap_w, self._rsmi, self._idx, self.path = None, None, None, local_rank, None
        try:
            lib = ctypes.CDLL("/opt/rocm/lib/librocm_smi64.so")
            if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                v = ctypes.c_uint64(0)
                if lib.rsmi_dev_current_socket_power_get(ctypes.c_uint32(local_rank), ctypes.byref(v)) == 0 and v.value > 0:
                    self._rsmi, self.kind = lib, "rsmi_dev_current_socket_power_get (librocm_smi64)"
                    cap = ctypes.c_uint64(0)
                    if lib.rsmi_dev_power_cap_get(ctypes.c_uint32(local_rank), ctypes.c_uint32(0), ctypes.byref(cap)) == 0:
                        self.cap_w = cap.value / 1e6
        except Exception:
            self._rsmi = None
        if self._rsmi is None:
            cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
            if cards:
                self.path = cards[min(local_rank, len(cards) - 1)]
                self.kind = self.path + " (hwmon: a slow moving average)"
                try:
                    self.cap_w = int(open(os.path.join(os.path.dirname(self.path), "power1_cap")).read()) / 1e6
                except Exception:
                    self.cap_w = None
        self.samples, self.clocks, self._stop, self._thread = [], [], False, None

    def _read_sclk(self):
        """current shader clock in MHz (rsmi_dev_gpu_clk_freq_get, RSMI_CLK_TYPE_SYS) or None: what the power management leaves of the clock while the
        package sits at its cap -- the throttle evidence next to the watts"""
        if self._rsmi is None:
            return None
        import ctypes

        class Freqs(ctypes.Structure):  # rsmi_frequencies_t
            _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * 33)]

        f = Freqs()
        if self._rsmi.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(self._idx), ctypes.c_uint32(0), ctypes.byref(f)) != 0 or f.num_supported == 0:
            return None
        cur = f.frequency[min(f.current, f.num_supported - 1, 32)]
        return cur / 1e6 if cur > 0 else None

    def _read(self):
        if self._rsmi is not None:
            import ctypes
            v = ctypes.c_uint64(0)
            return v.value / 1e6 if self._rsmi.rsmi_dev_current_socket_power_get(ctypes.c_uint32(self._idx), ctypes.byref(v)) == 0 else None
        return int(open(self.path).read()) / 1e6

    def _run(self):
        while not self._stop:
            try:
                w = self._read()
                if w:
                    self.samples.append(w)
                if len(self.samples) % 4 == 1:  # (every 200 ms)
                    c = self._read_sclk()
                    if c:
                        self.clocks.append(c)
            except Exception:
                pass
            time.sleep(0.05)

    def start(self):
        if self.kind is not None:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join()
        if not self.samples:
            return None
        s = sorted(self.samples)
        out = dict(mean_w=round(sum(s) / len(s), 1), median_w=round(s[len(s) // 2], 1), max_w=round(s[-1], 1), cap_w=self.cap_w, samples=len(s),
                   source=self.kind)
        if self.cap_w:  # how much of the timed region ran AT the cap: the convolutions do, the bandwidth-bound passes between them do not
            out["frac_samples_ge_95pct_of_cap"] = round(sum(1 for w in s if w >= 0.95 * self.cap_w) / len(s), 3)
            out["frac_samples_ge_90pct_of_cap"] = round(sum(1 for w in s if w >= 0.90 * self.cap_w) / len(s), 3)
        if self.clocks:
            c = sorted(self.clocks)
            out["sclk_mhz"] = dict(mean=round(sum(c) / len(c)), min=round(c[0]), max=round(c[-1]), samples=len(c), source="rsmi_dev_gpu_clk_freq_get(RSMI_CLK_TYPE_SYS)")
        return out


C2 = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(64, 128, 256), attention_levels=(False, False, False),
          num_res_blocks=2, num_head_channels=(0, 0, 256), norm_num_groups=32)


def rerandomize_zero_params(state_dict, seed=1234, std=0.05):
    """A freshly constructed UNet is the zero function (zero-initialised conv2 / out conv): give every all-zero parameter
    N(0, std) values from a fixed seed so the benchmark does real arithmetic on non-degenerate data."""
    g = torch.Generator().manual_seed(seed)
    for _, p in state_dict.items():
        if p.is_floating_point() and p.numel() > 0 and p.abs().max() == 0:
            p.copy_(torch.randn(p.shape, generator=g) * std)
    return state_dict


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_hbm_traffic_current.json")
# kernel symbol (prefix) behind each label of ops' per-launch profile, as rocprofv3 prints it (T = unsigned short / float)
DMA_CFG_TEMPLATE = {11: "8, 2, 1, 4, 3, 4", 14: "4, 4, 1, 2, 3, 4", 15: "8, 1, 2, 2, 3, 4", 16: "16, 2, 1, 4, 3, 4", 17: "4, 4, 1, 2, 2, 4",
                    18: "8, 4, 1, 2, 3, 4", 19: "8, 4, 1, 2, 3, 8"}


def kernel_source_sha() -> str:
    """Fingerprint of the HIP sources: PMC counters cannot be collected inside the timed run, so `roofline.traffic` comes from a committed
    rocprofv3 --pmc pass of this same command (tools/pmc_traffic.sh) -- valid only while the kernels it measured are the ones built now."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "generativemodels_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")):  # every translation unit whose kernels the PMC file carries rows for, and every header
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def measured_hbm_traffic(label: str, dtype) -> dict:
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this same bench command).  Correction as the guide's HBM section asks -- calibrated on known byte
    counts in this kernel's own access pattern (profiles/r03_fetch_size_calibration.json): FETCH_SIZE counts 64 B per request, so the
    64-byte activation pieces these convolutions fetch are counted exactly while a wide streaming read is counted at half; WRITE_SIZE is
    exact.  `traffic` = FETCH_SIZE + WRITE_SIZE for this kernel, `traffic_bracket` = [F + W, 2F + W] (the upper end would hold if every
    read were a wide one).  The file is stamped with the fingerprint of the kernel sources it measured: when the
    sources have changed since, the figure is stale and `traffic` is null (with the reason) instead of a number from another kernel."""
    try:
        cfg = int(label.split("cfg")[1].rstrip(">"))
        doc = json.load(open(TRAFFIC_FILE))
    except Exception:
        return dict(traffic=None, traffic_note="no committed PMC pass (profiles/pmc_hbm_traffic_current.json)")
    if doc.get("source_sha") != kernel_source_sha():
        return dict(traffic=None, traffic_note=f"stale: the committed PMC pass measured kernel sources {doc.get('source_sha')}, "
                                               f"this build is {kernel_source_sha()} -- re-run tools/pmc_traffic.sh")
    elem = "unsigned short" if dtype == torch.bfloat16 else "float"
    sym = f"conv_dma_kernel<{elem}, {DMA_CFG_TEMPLATE[cfg]}, false>" if cfg in DMA_CFG_TEMPLATE else None  # (last argument: no fused prologue)
    for r in doc.get("rows", []):
        if sym is not None and sym in r["kernel"]:
            return dict(traffic=round(r["hbm_mb_per_launch_corrected"] * 1e6), traffic_unit="bytes/launch (avg over the same launches)",
                        traffic_bracket=[round(v * 1e6) for v in r.get("hbm_mb_per_launch_bracket", [])] or None,
                        traffic_source=os.path.relpath(TRAFFIC_FILE, ROOT), traffic_launches_sampled=r["launches"], traffic_git_head=doc.get("git_head"))
    return dict(traffic=None, traffic_note=f"the committed PMC pass holds no row for {sym}")


def needs_self_launch(gpus: int, env) -> bool:
    """True when this process was started as plain `python bench.py --gpus N` (no launcher environment) and has to become N ranks itself:
    N > 1, or GM_BENCH_SELF_LAUNCH=1 (the GPU suite forces the path at N = 1, the only size a 1-GPU box can run)."""
    if "WORLD_SIZE" in env or "RANK" in env:
        return False
    return gpus > 1 or env.get("GM_BENCH_SELF_LAUNCH", "0") == "1"


def self_launch_command(gpus: int, argv, port: int | None = None):
    """The command (and environment additions) `python bench.py --gpus N ...` re-executes itself as: one rank per GPU of THIS node under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), the reference's own multi-GPU entry
    (tutorials/generative/distributed_training/ddpm_training_ddp.py:117-127,199 is launched the same way by torchrun)."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *argv]
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),  # dmabuf IPC: the only form the host driver supports
           "GM_BENCH_SELF_LAUNCHED": "1", "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(gpus, 1))))}
    return cmd, env


def self_launch(gpus: int, argv) -> int:
    import subprocess

    if torch.cuda.is_available() and torch.cuda.device_count() < gpus:
        raise SystemExit(f"--gpus {gpus} but this node shows {torch.cuda.device_count()} GPU(s)")
    cmd, extra = self_launch_command(gpus, argv)
    env = dict(os.environ)
    env.pop("GM_BENCH_SELF_LAUNCH", None)
    env.update(extra)
    return subprocess.run(cmd, env=env).returncode  # rank 0 of the child job prints the ONE JSON line on this process's stdout


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sampling chains (volumes) per GPU")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=128, help="volume edge (128 = the BASELINE config)")
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", type=int, default=0, help="replay the UNet forward from a HIP graph (measured slower than eager launches on ROCm 7.2 at this size: 18.7-19.5 vs 14.5-14.8 ms per iteration, round 3)")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "small", "off"])
    args = ap.parse_args()

    if needs_self_launch(args.gpus, os.environ):  # `python bench.py --gpus N` as the driver calls it: become N ranks under torch.distributed.run
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: generativemodels_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):  # launched by torch.distributed.run (also at N = 1: the
        import torch.distributed as dist  # RCCL over xGMI; used only to bracket the timing       # RCCL path then runs with one rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from generativemodels_amd import ops
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.nets import DiffusionModelUNet
    from generativemodels_amd.networks.schedulers import DDIMScheduler

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.set_grad_enabled(False)  # sampling only: with gradients enabled a network's forward takes its differentiable (training) path
    torch.manual_seed(0)
    model = DiffusionModelUNet(**C2).eval()
    sd = rerandomize_zero_params({k: v.clone() for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model = model.to(dev, dtype)
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(args.inference_steps)
    inferer = DiffusionInferer(sched, use_hip_graph=bool(args.graph))
    shape = (1, 1, args.size, args.size, args.size)
    noise = torch.randn(shape, generator=torch.Generator().manual_seed(7 + rank)).to(dev, dtype)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = inferer.sample(noise, model, sched, verbose=False)
    sync()
    power = PowerSampler(local_rank) if rank == 0 else None
    if power is not None:
        power.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = inferer.sample(noise, model, sched, verbose=False)
    sync()
    elapsed = time.perf_counter() - t0
    power_stats = power.stop() if power is not None else None
    if dist is not None:  # max over ranks
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(out.float()).all().item())

    # ---- per-kernel roofline: one instrumented eager forward (HIP events on the launch stream around every launch) ---------
    roof = None
    fwd_ms = None
    eps_gpu = None
    breakdown = {}
    if rank == 0:
        tt = torch.tensor([500.0], device=dev)
        eps_gpu = model(noise, tt).float().cpu()  # warm; kept: compared with the CPU oracle's forward of the same input below
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            model(noise, tt)
        e1.record()
        torch.cuda.synchronize()
        fwd_ms = e0.elapsed_time(e1) / 3
        ops.start_profile()
        model(noise, tt)
        rec = ops.stop_profile()
        for name, meta, ms in rec:
            b = breakdown.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            b["launches"] += 1
            b["ms"] += ms
            b["flops"] += meta["flops"]
            b["bytes"] += meta["bytes"]
        conv = {k: v for k, v in breakdown.items() if k.startswith("conv_igemm")}
        if conv:
            name, b = max(conv.items(), key=lambda kv: kv[1]["ms"])
            achieved = b["flops"] / (b["ms"] * 1e-3) / 1e12
            roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 2), peak=MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else 157.3,
                        unit="TFLOP/s", frac=round(achieved / (MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else 157.3), 4), traffic=None,
                        launches=b["launches"], avg_launch_ms=round(b["ms"] / b["launches"], 4),
                        algorithmic_gflop_per_launch=round(b["flops"] / b["launches"] / 1e9, 2),
                        algorithmic_mb_per_launch=round(b["bytes"] / b["launches"] / 1e6, 2))

        if roof is not None:
            roof.update(measured_hbm_traffic(roof["kernel"], dtype))
            if dtype == torch.bfloat16:
                key = "one ds_read_b128 per 32x32x16 MFMA (this kernel's tap loop)"
                best_lds, regs = MFMA_BF16_SUSTAINED_TFLOPS["0.75 ds_read_b128 per MFMA, 2 work-groups per CU (the best LDS-fed loop measured)"], MFMA_BF16_SUSTAINED_TFLOPS["operands in registers, no memory traffic"]
                roof.update(sustained_peak=MFMA_BF16_SUSTAINED_TFLOPS[key], frac_of_sustained=round(roof["achieved"] / MFMA_BF16_SUSTAINED_TFLOPS[key], 4),
                            frac_of_best_lds_fed_ceiling=round(roof["achieved"] / best_lds, 4), frac_of_register_resident_ceiling=round(roof["achieved"] / regs, 4),
                            sustained_peak_note="bf16 MFMA rate the chip sustains at its package power cap on random operands (tools/mfma_power.hip, "
                                                "profiles/r04_mfma_power_ceiling.txt): " + json.dumps(MFMA_BF16_SUSTAINED_TFLOPS) +
                                                "; the convolutions of this benchmark run at the cap (1 357-1 395 W per launch loop, profiles/r05_clock_energy.json), the "
                                                "bandwidth-bound passes between them (gn_apply, ~13 % of an iteration) do not: package_power_w.mean_w is the time-weighted mix and "
                                                "frac_samples_ge_95pct_of_cap the share of 50-ms samples at the cap; `frac` is against the nominal dense peak")

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----------------------------------------------------
    # (1) thread-count sweep at 1/8 of the voxels (oneDNN oversubscribes on many-core hosts: 256 threads measured 2x slower than 8 in
    # round 1), (2) with the best count: DDIM steps (UNet forward + scheduler step) at full size at t = 980, 500, 20 -- median;
    # (3) the t = 500 prediction is compared with the GPU forward of the same noise volume.
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline != "off":
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import restatement as R  # test infrastructure: the CPU statement of the reference algorithm (checker / baseline only; never on the product path)

        cores = os.cpu_count() or 1
        sd32 = {k: v.float() for k, v in sd.items()}
        # kind = "port": the reference is Python and does not travel to the GPU box in any form; the restatement is pinned to it by the golden
        # fixtures (tests/test_oracle_golden.py) -- at THIS size by tests/golden/c2_fullsize_ref.pt, compared again below
        cpu_forward = lambda xx, tt_: R.unet_forward(sd32, C2, xx, tt_)                        # noqa: E731
        cpu_step = lambda eps_, t_, xx: R.ddim_step(sched.alphas_cumprod, 1000, args.inference_steps, eps_, t_, xx, clip_sample=False)  # noqa: E731
        size = args.size if args.cpu_baseline == "full" else min(args.size, 48)
        x = torch.randn((1, 1, size, size, size), generator=torch.Generator().manual_seed(7))
        half = max(size // 2, 8)
        xs = x[..., :half, :half, :half].contiguous()
        sweep = {}
        with torch.no_grad():
            for nt in sorted({min(cores, c) for c in (8, 16, 32, 64, 128)}):  # (256 threads: 20x off the optimum on the r4 driver box, 46 s for nothing)
                torch.set_num_threads(nt)
                cpu_forward(xs[..., : half // 2, : half // 2, : half // 2].contiguous(), torch.tensor([500.0]))  # spin the pool up
                c0 = time.perf_counter()
                cpu_forward(xs, torch.tensor([500.0]))
                sweep[nt] = round(time.perf_counter() - c0, 3)
            best = min(sweep, key=sweep.get)
            torch.set_num_threads(best)
            times, eps500 = [], None
            for tstep in ((980, 500, 20) if args.cpu_baseline == "full" else (500,)):
                c0 = time.perf_counter()
                eps = cpu_forward(x, torch.tensor([float(tstep)]))
                cpu_step(eps, tstep, x)
                times.append(time.perf_counter() - c0)
                if tstep == 500:
                    eps500 = eps
        cpu_s = sorted(times)[len(times) // 2]
        scale = (args.size / size) ** 3
        cpu = dict(value=round(1.0 / (cpu_s * scale * args.inference_steps), 8), unit="volumes/s", cores=best, kind="port",
                   host_cores=cores, seconds_per_step=round(cpu_s * scale, 3), step_seconds=[round(v, 2) for v in times],
                   thread_sweep_seconds_at_half_edge=sweep,
                   sample=f"median of {len(times)} DDIM steps (UNet forward + scheduler step, fp32, torch-CPU oracle, t = 980/500/20) at 1x1x{size}^3 on "
                          f"{best} threads (best of a sweep over {sorted(sweep)} at 1x1x{half}^3)"
                          + ("" if size == args.size else f", scaled x{scale:.0f} to {args.size}^3 by voxel count") + f", x{args.inference_steps} steps",
                   port_note="kind=port: oracle/restatement.py, the reference algorithm restated on torch-CPU (the reference is Python and does not travel to the "
                             "GPU box); in the build container (8 cores) it runs the C2 forward at 1x1x64^3 in 1.09x the time of the unmodified reference module "
                             "(tools/oracle_vs_reference_time.py, profiles/r02_oracle_vs_reference_time.json)")
        fx_path = os.path.join(ROOT, "tests", "golden", "c2_fullsize_ref.pt")  # outputs of the unmodified reference at this very size (oracle/make_golden_c2_fullsize.py)
        if eps500 is not None and size == args.size == 128 and os.path.exists(fx_path):
            fx = torch.load(fx_path, weights_only=False)
            L = fx["lattice"]
            cpu.update(port_vs_reference_max_abs_diff=round((eps500[..., ::L, ::L, ::L] - fx["fp32"][500]["lattice"]).abs().max().item(), 8),
                       port_vs_reference_note="this run's t = 500 oracle output against the reference's own (every-4th-voxel lattice of tests/golden/c2_fullsize_ref.pt)")
        if eps500 is not None and size == args.size and eps_gpu is not None and args.dtype == "bf16":
            err = (eps_gpu.reshape(eps500.shape) - eps500).abs()
            sigma = eps500.std().item()
            cpu.update(max_abs_err_vs_gpu=round(err.max().item(), 5), mean_abs_err_vs_gpu=round(err.mean().item(), 6), oracle_output_sigma=round(sigma, 4),
                       gpu_matches_oracle=bool(err.mean().item() <= 2e-2 * sigma and err.max().item() <= 0.2 * sigma),
                       parity_bar="bf16 GPU forward vs fp32 oracle at t = 500 on the benchmark's own noise volume: mean <= 2e-2 sigma, max <= 0.2 sigma")

    if rank == 0:
        vol_s = world * args.steps / elapsed
        line = {
            "metric": "3D volumes/sec (DDIM 50-step sample, 1x128^3)", "value": round(vol_s, 5), "unit": "volumes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (Gaussian noise volumes, random-init weights)",
            "config": {"workload": f"C2: 3D DiffusionModelUNet(64,128,256; 2 res blocks; mid-block attention 1 head x 256) DDIM-{args.inference_steps} "
                                   f"sampling of 1x1x{args.size}^3 volumes, 1 volume per GPU per step",
                       "volumes_per_gpu_per_step": 1, "inference_steps": args.inference_steps,
                       "parallelism": f"{world} independent replicas (batch-sharded, no data-path collective)", "hip_graph": bool(args.graph),
                       "process_group": None if dist is None else f"nccl (RCCL), world_size {world}",
                       "self_launched": os.environ.get("GM_BENCH_SELF_LAUNCHED") == "1"},
            "unet_forward_ms": None if fwd_ms is None else round(fwd_ms, 3),
            "ms_per_ddim_iteration": round(1e3 * elapsed / args.steps / args.inference_steps, 3),
            "output_finite": finite,
            "roofline": roof, "package_power_w": power_stats,
            "joules_per_volume": None if not power_stats else round(power_stats["mean_w"] * elapsed / args.steps, 1),  # what bounds this loop: it runs at the power cap
            "cpu_baseline": cpu,
            "speedup_vs_cpu": None if cpu is None else round(vol_s / cpu["value"], 1),
            # per kernel label: algorithmic FLOP and HBM rates of one forward; frac_mfma / frac_hbm = those rates over the nominal peaks (the larger one names the bound)
            "kernel_breakdown_ms": {k: dict(launches=v["launches"], ms=round(v["ms"], 3),
                                            tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2), gbs=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                                            frac_mfma=round(v["flops"] / max(v["ms"], 1e-9) / 1e9 / (MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else 157.3), 3),
                                            frac_hbm=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 3))
                                    for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"])},
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
