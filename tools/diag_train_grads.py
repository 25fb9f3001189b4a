import sys, math, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import restatement as R
from test_gpu_backward import COND_TRAIN_CASES, _rand
from generativemodels_amd.networks.nets import DiffusionModelUNet
DEV = "cuda"
for case in ("cond3d",):
    c = COND_TRAIN_CASES[case]; cfg = c["cfg"]
    torch.manual_seed(13)
    model = DiffusionModelUNet(**cfg); R.derandomize_zeros(model, seed=8)
    x, ctx = _rand(c["shape"], 411), _rand(c["context"], 412)
    t = torch.tensor([40, 731])
    labels = None if c["class_labels"] is None else torch.tensor(c["class_labels"])
    target = _rand((c["shape"][0], cfg["out_channels"], *c["shape"][2:]), 413)
    sd = {k_: v_.detach().double().requires_grad_(True) for k_, v_ in model.state_dict().items()}
    y_ref = R.unet_forward(sd, cfg, x.double(), t, ctx.double(), labels)
    F.mse_loss(y_ref, target.double()).backward()
    model = model.to(DEV)
    y = model.forward_train(x.to(DEV), t.to(DEV), context=ctx.to(DEV), class_labels=None if labels is None else labels.to(DEV))
    print("fwd err", float((y.detach().cpu().double() - y_ref.detach()).abs().max()))
    F.mse_loss(y, target.to(DEV)).backward()
    rows = []
    for name, p in model.named_parameters():
        if p.grad is None: continue
        w = sd[name].grad
        err = float((p.grad.detach().cpu().double() - w).abs().max()); sc = float(w.abs().max())
        rows.append((err / max(sc, 1e-12), err, sc, name))
    for i, (rel, err, sc, name) in enumerate(rows):
        flag = "BAD" if rel > 1e-3 else "ok "
        print(f"{i:3d} {flag} rel {rel:.2e} err {err:.2e} scale {sc:.2e} {name}")
