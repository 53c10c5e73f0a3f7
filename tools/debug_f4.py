import os, sys
os.environ["PNP_WINO_MIN_BLOCKS"]="1"; os.environ["PNP_WINO_F4_MIN_CIN"]="128"
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from dt4image_restoration_amd import synthetic, weights
from dt4image_restoration_amd.engine import PnPEngine
from oracle import pnp_oracle as O
n,h,w=2,128,128
sd_np=weights.generate_unet_weights(0,"unit_gain")
e=PnPEngine(n,h,w,keep_stages=True); e.load_weights(sd_np)
print(e.conv_algorithms())
sd=O.torch_weights(sd_np)
x=(torch.from_numpy(synthetic.hash_uniform(19,h*100+w,n*h*w).reshape(n,1,h,w))+1)*0.5
sigma=torch.linspace(5,50,n)/255.0
got=e.denoise(x.cuda(),sigma.cuda())
nm=torch.ones(n,1,h,w)*sigma.view(n,1,1,1)
ref_raw,stages=O.unet_forward(sd,torch.cat([x,nm],1),return_stages=True)
for which,(name,ref) in enumerate(stages.items()):
    a=e.read_stage(which).cpu()
    d=(a-ref).abs()
    print(name, tuple(ref.shape), "max err", float(d.max()))
    if float(d.max())>1e-3:
        pc=d.amax(dim=(0,2,3)); print(" per-channel err (first 32):", np.round(pc[:32].numpy(),3))
        pp=d.amax(dim=(0,1)); print(" per-pixel err top-left 8x8:\n", np.round(pp[:8,:8].numpy(),2))
        print(" ref[0,:8,0,0]", ref[0,:8,0,0].numpy(), "\n got[0,:8,0,0]", a[0,:8,0,0].numpy())
        print(" ref[0,0,0,:8]", ref[0,0,0,:8].numpy(), "\n got[0,0,0,:8]", a[0,0,0,:8].numpy())
        break
