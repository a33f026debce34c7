import torch, time
dev=torch.device("cuda:0")
n=100000*65536
a=torch.empty(n,dtype=torch.uint8,device=dev); b=torch.empty(n,dtype=torch.uint8,device=dev)
a.random_(0,255)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
ts=[]
for _ in range(10):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"torch copy_ of {n/1e9:.2f} GB: min {min(ts):.3f} ms = {n/min(ts)/1e6:.0f} GB/s per direction ({2*n/min(ts)/1e6:.0f} GB/s read+write)")
a4=a.view(torch.int32); b4=b.view(torch.int32)
ts=[]
for _ in range(10):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); b4.copy_(a4); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"as int32: min {min(ts):.3f} ms = {n/min(ts)/1e6:.0f} GB/s per direction")
