import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e-3
for shape in ((4,32,1024,1024),(8,32,1024,1024),(4,128,256,256)):
    x=torch.randn(shape,device='cuda'); y=torch.empty_like(x); z=torch.randn(shape,device='cuda')
    b=x.numel()*4
    print(shape,'copy %.0f GB/s'%(2*b/t(lambda: y.copy_(x))/1e9),'add(read2,write1) %.0f GB/s'%(3*b/t(lambda: torch.add(x,z,out=y))/1e9),'sum(read) %.0f GB/s'%(b/t(lambda: x.sum())/1e9), 'fill(write) %.0f GB/s'%(b/t(lambda: y.fill_(1.0))/1e9))
