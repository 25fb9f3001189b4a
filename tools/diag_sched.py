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
