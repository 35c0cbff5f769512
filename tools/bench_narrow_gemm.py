import sys, torch
sys.path.insert(0, '/root/repo')
from transoar_amd import gemm
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(it+1)]
    ev[0].record()
    for i in range(it): fn(); ev[i+1].record()
    torch.cuda.synchronize()
    ts=sorted(ev[i].elapsed_time(ev[i+1]) for i in range(it)); return ts[len(ts)//2]
for (m,k,n) in [(1638400,48,48),(1638400,192,48),(1638400,144,48),(204800,384,96),(204800,96,96),(1638400,48,144),(1638400,48,192)]:
    x=torch.randn(m,k,device='cuda').bfloat16(); w=torch.randn(n,k,device='cuda').bfloat16(); b=torch.randn(n,device='cuda')
    ms=t(lambda: gemm.linear_nt(x,w,b))
    print(m,k,n,"%.4f ms  %.0f GB/s"%(ms,(m*(k+n)*2)/ms/1e6))
