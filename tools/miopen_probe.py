import time, torch, torch.nn as nn, torch.nn.functional as F
dev="cuda"
torch.manual_seed(0)
N=2450
enc=nn.Sequential(nn.Conv2d(3,32,4,2),nn.ELU(),nn.Conv2d(32,64,4,2),nn.ELU(),nn.Conv2d(64,128,4,2),nn.ELU(),nn.Conv2d(128,256,4,2),nn.ELU(),nn.Flatten()).to(dev)
dec=nn.Sequential(nn.Linear(230,1024),nn.Unflatten(1,(1024,1,1)),nn.ConvTranspose2d(1024,128,5,2),nn.ELU(),nn.ConvTranspose2d(128,64,5,2),nn.ELU(),nn.ConvTranspose2d(64,32,6,2),nn.ELU(),nn.ConvTranspose2d(32,3,6,2)).to(dev)
x=torch.randn(N,3,64,64,device=dev); f=torch.randn(N,230,device=dev,requires_grad=True)
for it in range(4):
    torch.cuda.synchronize(); t0=time.time()
    e=enc(x); o=dec(f); loss=(e.sum()+((o-x)**2).sum()); loss.backward()
    torch.cuda.synchronize(); print(f"iter {it}: {1e3*(time.time()-t0):.1f} ms", e.shape, o.shape, flush=True)
