"""GPU: the three attention-backward paths (ops.attention_backward = fused fp32-MFMA flash kernels; the fp32 composed path of autograd.py;
ops.attention_backward_bf16 = bf16-MFMA score pass + weight-gradient / 1x1 kernels) at training shapes.   usage: python tools/cmp_attention_backward.py"""
import sys, json, math, torch
sys.path.insert(0, "/root/repo")
from generativemodels_amd import ops, autograd as A
def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(reps+1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i+1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i+1]) for i in range(reps))
for b,l,h,dh in [(1,4096,1,128),(1,512,1,256),(1,8192,1,64),(2,1024,4,32),(1,32768,1,256)]:
    c=h*dh
    q,k,v,go=(torch.randn((b,l,c),device="cuda").bfloat16() for _ in range(4))
    sc=1/math.sqrt(dh)
    o=ops.attention(q,k,v,h,sc)
    t_f=timeit(lambda: ops.attention_backward(q,k,v,o,go,h,sc)) if l <= 8192 else None
    t_b=timeit(lambda: ops.attention_backward_bf16(q,k,v,o,go,h,sc))
    ops.start_profile(); ops.attention_backward_bf16(q,k,v,o,go,h,sc); rec=ops.stop_profile()
    parts={}
    for name,meta,ms in rec: parts[name]=round(parts.get(name,0.0)+ms,3)
    t_c=None
    if l <= 8192:
        saved=(ops.ATTENTION_BWD_HEAD_DIMS, A.ATTENTION_BWD_BF16_MIN_TOKENS)
        def composed():
            ops.ATTENTION_BWD_HEAD_DIMS=(); A.ATTENTION_BWD_BF16_MIN_TOKENS=1<<30
            qq,kk,vv=(t.clone().requires_grad_(True) for t in (q,k,v))
            oo=A.attention(qq,kk,vv,h,sc); oo.backward(go)
            ops.ATTENTION_BWD_HEAD_DIMS, A.ATTENTION_BWD_BF16_MIN_TOKENS=saved
        def fwd_only():
            qq,kk,vv=(t.clone().requires_grad_(True) for t in (q,k,v))
            A.attention(qq,kk,vv,h,sc)
        t_c=timeit(composed)-timeit(fwd_only)
    flops=10.0*b*h*l*l*dh
    print(json.dumps(dict(B=b,L=l,H=h,dh=dh,fused_fp32_ms=None if t_f is None else round(t_f,3),composed_fp32_ms=None if t_c is None else round(t_c,3),
                          bf16_mfma_ms=round(t_b,3),bf16_tflops_of_5_gemms=round(flops/t_b/1e9,1),bf16_parts_ms=parts)),flush=True)
