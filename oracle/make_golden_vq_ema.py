"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).  Golden outputs of the UNMODIFIED reference EMAQuantizer in train()
mode -- the EMA codebook update of generative/networks/layers/vector_quantizer.py:161-188 -- for tests/golden/vq_ema.pt:
two consecutive training forwards (state carries over), the buffers after each, and the input gradient of (quantized, loss)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    gen = load_reference()
    from generative.networks.layers.vector_quantizer import EMAQuantizer

    cases = {}
    for name, (sd_, k, d, shape, decay, eps) in {"q3d": (3, 16, 8, (2, 8, 4, 4, 4), 0.9, 1e-5), "q2d": (2, 5, 3, (1, 3, 6, 7), 0.5, 1e-3)}.items():
        torch.manual_seed(3)
        layer = EMAQuantizer(spatial_dims=sd_, num_embeddings=k, embedding_dim=d, commitment_cost=0.25, decay=decay, epsilon=eps).train()
        init = {n: v.clone() for n, v in layer.state_dict().items()}
        steps = []
        for s in range(2):
            x = torch.randn(shape, generator=torch.Generator().manual_seed(40 + s)).requires_grad_(True)
            q, loss, idx = layer(x)
            gq = torch.randn(shape, generator=torch.Generator().manual_seed(50 + s))
            ((q * gq).sum() + 3.0 * loss).backward()
            steps.append(dict(x=x.detach().clone(), quantized=q.detach().clone(), loss=loss.detach().clone(), indices=idx.clone(), gq=gq,
                              dx=x.grad.clone(), state={n: v.clone() for n, v in layer.state_dict().items()}))
        cases[name] = dict(args=dict(spatial_dims=sd_, num_embeddings=k, embedding_dim=d, commitment_cost=0.25, decay=decay, epsilon=eps),
                           init=init, steps=steps)
        print(name, [float(s["loss"]) for s in steps])
    torch.save(dict(kind="vq_ema", cases=cases), os.path.join(OUT, "vq_ema.pt"))


if __name__ == "__main__":
    main()
