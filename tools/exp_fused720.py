import os, sys
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/phantom-fhe_amd"]
import torch
import phantom_fhe_amd as P
n=1<<16
primes=[int(p) for p in P.coeff_modulus_create(n,[60]+[50]*44+[60]*15)]
dev=torch.device("cuda:0")
ctx=P.PhantomContext(16,primes,15,device=dev)
g=torch.Generator(device=dev); g.manual_seed(1)
polys=torch.randint(0,1<<49,(16,45,n),dtype=torch.int64,device=dev,generator=g)
ref=None
def timed(fn,steps=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/steps*1e3
DEFAULT=1|32|64|2048|4096
for name,var,extra in (("product plan (64 x 1024, batched contiguous pass)",DEFAULT,{}),
                       ("two launches, 256 x 256 (plan 3/4)",1|32|64|2048,{}),
                       ("ONE launch, L2 hand-off, lag 2",1|16|64|512,{3:2}),
                       ("ONE launch, L2 hand-off, lag 1",1|16|64|512,{3:1}),
                       ("ONE launch, L2 hand-off, lag 4",1|16|64|512,{3:4})):
    P.set_tuning(0,var)
    for k,v in extra.items(): P.set_tuning(k,v)
    x=polys.clone()
    ctx.nwt_2d_radix8_forward_inplace_batched(x,45,0,16,45*n)
    chk=int(x.sum().item())
    if ref is None: ref=chk
    f=min(timed(lambda: ctx.nwt_2d_radix8_forward_inplace_batched(x,45,0,16,45*n)) for _ in range(3))
    print(f"{f:8.1f} us per 720-limb step  frac {16.0*n*45*16/(f*1e-6)/8e12:.3f}  same words {chk==ref}   {name}",flush=True)
