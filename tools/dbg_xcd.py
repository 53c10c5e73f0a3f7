"""GPU box: the per-XCD persistent data-fidelity kernel (PNP_FFT_XCD=1) against the three-launch path, call after call:
    python tools/dbg_xcd.py N H [stop]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from dt4image_restoration_amd import synthetic
from dt4image_restoration_amd.engine import PnPEngine
n,h=int(sys.argv[1]),int(sys.argv[2]); w=h
d = synthetic.make_problem(n, h, w, accel=4.0, seed=71)
masks = torch.from_numpy(np.asarray(d["mask"])).reshape(h, w).bool()
y0 = torch.view_as_complex(torch.from_numpy(d["y0"])); x0 = torch.view_as_complex(torch.from_numpy(d["x0"]))
xd = torch.clamp(x0.real + 0.05 * torch.from_numpy(synthetic.hash_uniform(11, 1, n * h * w).reshape(n, 1, h, w)), 0, 1)
u0 = 0.1 * torch.view_as_complex(torch.from_numpy(synthetic.hash_uniform(11, 2, 2 * n * h * w).reshape(n, 1, h, w, 2).copy()))
mu = torch.linspace(0.05, 0.6, n)
tact = torch.zeros(n)
if len(sys.argv) > 3: tact[1::3] = 0.9
outs={}
for name in ("three","xcd"):
    if name=="xcd": os.environ["PNP_FFT_XCD"]="1"
    else: os.environ.pop("PNP_FFT_XCD",None)
    e = PnPEngine(n,h,w)
    e.reset(x0.cuda(), y0.cuda(), masks.cuda())
    res=[]
    for rep in range(6):
        xg, ug = xd.cuda(), u0.cuda().clone()
        zg = torch.full_like(ug, 7.0)
        e.prox_dual(xg, zg, ug, mu.cuda(), tact.cuda())
        torch.cuda.synchronize()
        res.append((torch.view_as_real(zg).clone(), torch.view_as_real(ug).clone()))
    outs[name]=res
    for i,r in enumerate(res[1:]):
        dz=(r[0]-res[0][0]).abs(); 
        bad=[int(k) for k in range(n) if dz[k].max()>0]
        print(name,"rep",i+1,"slices differing from rep 0:",bad, float(dz.max()))
dz=(outs["xcd"][0][0]-outs["three"][0][0]).abs()
print("xcd vs three: slices differing", [int(k) for k in range(n) if dz[k].max()>0], float(dz.max()))
k=[int(k) for k in range(n) if dz[k].max()>0]
if k:
    m=dz[k[0],0]; idx=(m.sum(-1)>0).nonzero(); print("slice",k[0],"bad pixels",len(idx),"rows",idx[:,0].unique()[:20].tolist(),"cols",idx[:,1].unique()[:20].tolist())
