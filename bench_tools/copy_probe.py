import sys
sys.path.insert(0,'/root/repo/swift-homomorphic-encryption_amd')
import torch, heamd
x=torch.randint(0,1<<62,(4096,4,8192),dtype=torch.int64,device='cuda'); y=torch.empty_like(x)
for nt in (False,True,False,True):
    for _ in range(5): heamd.stream_copy(x,y,nt)
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(30): heamd.stream_copy(x,y,nt)
    b.record(); b.synchronize()
    print('nt' if nt else 'plain', 2*x.numel()*8/(a.elapsed_time(b)/30*1e-3)/1e9, 'GB/s')
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(30): y.copy_(x)
b.record(); b.synchronize()
print('torch copy_', 2*x.numel()*8/(a.elapsed_time(b)/30*1e-3)/1e9)
