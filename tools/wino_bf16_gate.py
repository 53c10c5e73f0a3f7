"""CPU gate (oracle only, round 4): bf16 mode on Winograd F(2x2,3x3) - transformed activations V = B^T d B rounded to bf16, transformed weights
U = G g G^T as two bf16 terms, f32 accumulate, f32 output transform.  PSNR offset to the f32 reference over a 53-iteration episode:
    python tools/wino_bf16_gate.py 512 8 2 53     ->  0.0033 dB max (profiles/r04_ablation.md); the error gate passes, the operand-bandwidth
estimate does not, so no kernel was built."""
import sys, time, numpy as np, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import synthetic, weights
from oracle import pnp_oracle as O
torch.set_num_threads(8)
size=int(sys.argv[1]); accel=float(sys.argv[2]); n=int(sys.argv[3]); iters=int(sys.argv[4]) if len(sys.argv)>4 else 53
bf=O._bf16
Bt=torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=torch.float32)
G=torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],dtype=torch.float64)
At=torch.tensor([[1,1,1,0],[0,1,-1,-1]],dtype=torch.float32)
def wino_conv(x,w,b,uterms,round_v=True):
    N,C,H,W=x.shape; K=w.shape[0]
    xp=F.pad(x,(1,1,1,1))
    # tiles of 4x4 stride 2 -> [N, C, th, tw, 4, 4]
    t=xp.unfold(2,4,2).unfold(3,4,2)
    V=torch.einsum('ij,nchwjk,lk->nchwil',Bt,t,Bt)       # B^T d B
    if round_v: V=bf(V)
    U=torch.einsum('ij,kcjl,ml->kcim',G,w.double(),G)   # [K,C,4,4] f64
    Uf=U.float()
    if uterms==2:
        hi=bf(Uf); Uq=hi+bf(Uf-hi)
    elif uterms==1: Uq=bf(Uf)
    else: Uq=Uf
    M=torch.einsum('nchwil,kcil->nkhwil',V,Uq)
    Y=torch.einsum('ai,nkhwil,bl->nkhwab',At,M,At)       # [N,K,th,tw,2,2]
    th,tw=Y.shape[2],Y.shape[3]
    y=Y.permute(0,1,2,4,3,5).reshape(N,K,2*th,2*tw)
    return y+b.view(1,-1,1,1)
class Plan:
    def __init__(self,uterms,round_v=True): self.u=uterms; self.rv=round_v
def stage(sd,prefix,x,plan):
    for j in range(3):
        w=sd[f"{prefix}.conv-{j}.conv2d.weight"]; b=sd[f"{prefix}.conv-{j}.conv2d.bias"]
        if plan is not None and w.shape[1]>=32:
            y=wino_conv(bf(x),w,b,plan.u,plan.rv)
        else:
            y=F.conv2d(x,w,b,padding=1)
        x=F.leaky_relu(y,0.2)
    return x
def unet(sd,x,plan):
    up=lambda a: F.interpolate(a,scale_factor=2,mode='bilinear',align_corners=True)
    x1=stage(sd,"inc.conv",x,plan)
    x2=stage(sd,"down1.mpconv.1",F.max_pool2d(x1,2),plan)
    x3=stage(sd,"down2.mpconv.1",F.max_pool2d(x2,2),plan)
    x4=stage(sd,"down3.mpconv.1",F.max_pool2d(x3,2),plan)
    x5=stage(sd,"down4.mpconv.1",F.max_pool2d(x4,2),plan)
    y=stage(sd,"up1.conv",torch.cat([x4,up(x5)],1),plan)
    y=stage(sd,"up2.conv",torch.cat([x3,up(y)],1),plan)
    y=stage(sd,"up3.conv",torch.cat([x2,up(y)],1),plan)
    y=stage(sd,"up4.conv",torch.cat([x1,up(y)],1),plan)
    return x[:,:1]+F.conv2d(y,sd["outc.conv.weight"],sd["outc.conv.bias"])
def episode(sd,data,mu,sg,iters,plan):
    st=O.reset(data); hist=[]
    nn=st["z"].shape[0]
    for t in range(iters):
        z,u,y0,mask=st["z"],st["u"],st["y0"],st["mask"]
        d=(z-u).real
        nm=torch.ones_like(d)*torch.from_numpy(sg[:,t]).view(nn,1,1,1)
        xa=torch.clamp(unet(sd,torch.cat([d,nm],1),plan),0,1)
        zf=O.fft2c(xa+u); m=torch.from_numpy(mu[:,t]).view(nn,1,1,1)
        zf=torch.where(mask,(m*zf+y0)/(1+m),zf)
        zn=O.ifft2c(zf); st["u"]=u+xa-zn; st["z"]=zn; st["x"]=xa
        hist.append(O.psnr(xa,st["gt"])[:,0])
    return torch.stack(hist,1)
sd=O.torch_weights(weights.generate_unet_weights(0,"unit_gain"))
data=synthetic.make_problem(n,size,size,accel=accel,sigma_n=10/255.,seed=1234)
mu,sg=synthetic.param_table(n,iters,seed=77)
with torch.no_grad():
    t0=time.time(); ref=episode(sd,data,mu,sg,iters,None); print("f32 direct",time.time()-t0,flush=True)
    # sanity: winograd in f32 == direct
    for name,plan in (("wino V bf16, U two-term",Plan(2)),):
        t0=time.time(); h=episode(sd,data,mu,sg,iters,plan)
        d=(h-ref).abs().max(0).values.numpy()
        marks=[0,5,9,19,29,39,min(49,iters-1),iters-1]
        print(f"{name:28s} {time.time()-t0:6.0f}s "+" ".join(f"{m+1}:{d[m]:.4f}" for m in marks)+f" max {d.max():.4f}",flush=True)
