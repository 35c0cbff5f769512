import torch
dev="cuda"
def t(f,n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
for n in (234000*384, 234000*1024):
    x=torch.randn(n,device=dev).to(torch.bfloat16)
    print(n, "bernoulli_ u8 %.3f"%t(lambda: torch.empty(n,dtype=torch.uint8,device=dev).bernoulli_(0.9)),
      "bernoulli_ bool %.3f"%t(lambda: torch.empty(n,dtype=torch.bool,device=dev).bernoulli_(0.9)),
      "rand bf16 < %.3f"%t(lambda: torch.rand(n,device=dev,dtype=torch.bfloat16)<0.9),
      "randint u8 %.3f"%t(lambda: torch.randint(0,256,(n,),device=dev,dtype=torch.uint8)),
      "native_dropout %.3f"%t(lambda: torch.ops.aten.native_dropout(x,0.1,True)),
      "random_ int32/4 %.3f"%t(lambda: torch.empty(n//4,dtype=torch.int32,device=dev).random_()),
      "random_ int64/8 %.3f"%t(lambda: torch.empty(n//8,dtype=torch.int64,device=dev).random_()))
