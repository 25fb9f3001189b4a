"""GPU diagnostic: where does the fused scheduler step differ from the reference golden vectors, and by how many ulps?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from _util import load_fixture
from generativemodels_amd.networks.schedulers import DDIMScheduler, DDPMScheduler

def ulps(a, b):
    ai = a.contiguous().view(torch.int32).long(); bi = b.contiguous().view(torch.int32).long()
    return (ai - bi).abs()

fx = load_fixture("schedulers")
mo, xs = fx["model_output"], fx["sample"]
tot = bad = 0
for sname, e in fx["tables"].items():
    ddim = DDIMScheduler(1000, schedule=sname, clip_sample=False, **e["kw"]); ddim.set_timesteps(50)
    for (pt, clip, t, eta), (prev, x0) in e["ddim"].items():
        ddim.prediction_type, ddim.clip_sample = pt, clip
        gen = torch.Generator().manual_seed(fx["noise_seed"])
        p2, x2 = ddim.step(mo.cuda(), t, xs.cuda(), eta=eta, generator=gen)
        up, ux = ulps(p2.cpu(), prev), ulps(x2.cpu(), x0)
        tot += 1
        if up.max() or ux.max():
            bad += 1
            print("ddim", sname, pt, clip, t, eta, "prev: n", int((up > 0).sum()), "max ulp", int(up.max()), "| x0: n", int((ux > 0).sum()), "max ulp", int(ux.max()))
    ddpm = DDPMScheduler(1000, schedule=sname, **e["kw"])
    for (pt, vt, t), (prev, x0) in e["ddpm"].items():
        ddpm.prediction_type, ddpm.variance_type = pt, vt
        m = fx["model_output2"] if vt.startswith("learned") else mo
        gen = torch.Generator().manual_seed(fx["noise_seed"])
        p2, x2 = ddpm.step(m.cuda(), t, xs.cuda(), generator=gen)
        up, ux = ulps(p2.cpu(), prev), ulps(x2.cpu(), x0)
        tot += 1
        if up.max() or ux.max():
            bad += 1
            print("ddpm", sname, pt, vt, t, "prev: n", int((up > 0).sum()), "max ulp", int(up.max()), "| x0: n", int((ux > 0).sum()), "max ulp", int(ux.max()))
print("cases", tot, "mismatching", bad)

# detailed look at one mismatching case: which host-side formula does the GPU result equal?
import numpy as np
e = fx["tables"]["linear_beta"]
d = DDIMScheduler(1000, schedule="linear_beta", clip_sample=False); d.set_timesteps(50)
d.prediction_type = "epsilon"
t = 980
prev_ref, x0_ref = e["ddim"][("epsilon", False, 980, 0.0)]
p2, x2 = d.step(mo.cuda(), t, xs.cuda())
a_t, a_prev = d._abar(t), d._abar(t - 20)
c_prev = np.float32((a_prev ** 0.5).item()); c_dir = np.float32(((1 - a_prev) ** 0.5).item())
x0n, mn = x0_ref.numpy().ravel(), mo.numpy().ravel()
plain = (c_prev * x0n).astype(np.float32) + (c_dir * mn).astype(np.float32)
fma1 = (np.float64(c_prev) * x0n.astype(np.float64) + (c_dir * mn).astype(np.float32).astype(np.float64)).astype(np.float32)
fma2 = ((c_prev * x0n).astype(np.float32).astype(np.float64) + np.float64(c_dir) * mn.astype(np.float64)).astype(np.float32)
g = p2.cpu().numpy().ravel(); r = prev_ref.numpy().ravel()
print("c_prev", c_prev, "c_dir", c_dir)
print("ref==plain", np.array_equal(r, plain), "gpu==plain", np.array_equal(g, plain), "gpu==fma1", np.array_equal(g, fma1), "gpu==fma2", np.array_equal(g, fma2),
      "ref==fma1", np.array_equal(r, fma1), "ref==fma2", np.array_equal(r, fma2))
idx = np.nonzero(g != r)[0][:5]
for i in idx:
    print(i, "x0", x0n[i], "m", mn[i], "gpu", g[i], "ref", r[i], "plain", plain[i], "fma1", fma1[i], "fma2", fma2[i])
