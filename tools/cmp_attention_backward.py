import sys, json, math, torch
sys.path.insert(0, "/root/repo")
from generativemodels_amd import ops, autograd as A
sys.path.insert(0, "/root/repo/tools")
def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(reps+1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i+1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i+1]) for i in range(reps))
for b,l,h,dh in [(1,4096,1,128),(1,512,1,256),(1,8192,1,64),(2,1024,4,32)]:
    c=h*dh
    q,k,v,go=(torch.randn((b,l,c),device="cuda").bfloat16() for _ in range(4))
    sc=1/math.sqrt(dh)
    o=ops.attention(q,k,v,h,sc)
    t_f=timeit(lambda: ops.attention_backward(q,k,v,o,go,h,sc))
    saved=ops.ATTENTION_BWD_HEAD_DIMS
    def composed():
        ops.ATTENTION_BWD_HEAD_DIMS=()
        qq,kk,vv=(t.clone().requires_grad_(True) for t in (q,k,v))
        oo=A.attention(qq,kk,vv,h,sc); oo.backward(go)
        ops.ATTENTION_BWD_HEAD_DIMS=saved
    def fwd_only():
        qq,kk,vv=(t.clone().requires_grad_(True) for t in (q,k,v))
        A.attention(qq,kk,vv,h,sc)
    t_c=timeit(composed)-timeit(fwd_only)
    print(json.dumps(dict(B=b,L=l,H=h,dh=dh,fused_ms=round(t_f,3),composed_ms=round(t_c,3))),flush=True)
