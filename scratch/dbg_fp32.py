import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'diffusion-model_amd')]
import numpy as np, torch
from isingdiff.keras_file import load_unet_spec
from isingdiff.model import ScoreModel
from oracle import ref_cpu
spec=load_unet_spec(os.path.join(ROOT,'weights/b_diffusion_model_2d_ising_realspace_unet.keras'))
ref=ref_cpu.RefModel(spec); gpu=ScoreModel(spec,0,'fp32')
B,L,T=4,20,1000
rs=np.random.RandomState(0)
x=rs.standard_normal((B,L,L,1)).astype(np.float32)
z=rs.standard_normal((T,B,L,L,1)).astype(np.float32)
betas=ref_cpu.beta_schedule(T)
for t in range(T-1,-1,-1):
    tn=np.float32(t)/np.float32(T)
    s,taps=ref.forward(x,tn,taps=True)
    g,gt=gpu.forward_taps(x,tn,precision='fp32')
    bad=False
    for i in range(10):
        a=gt[i].cpu().numpy(); b=taps[i]
        nb=np.count_nonzero(a.view(np.uint32)!=b.view(np.uint32))
        if nb:
            idx=np.argwhere(a.view(np.uint32)!=b.view(np.uint32))[:5]
            print('t',t,'layer',i,'nbad',nb,'of',a.size)
            for j in idx:
                j=tuple(j); print('   ',j,repr(a[j]),repr(b[j]), a[j].view(np.uint32)-b[j].view(np.uint32) if True else '')
            bad=True; break
    if bad:
        print('max|x|',np.abs(x).max())
        # check inputs to that layer equal
        break
    x=ref_cpu.update(x,s,z[t],betas[t])
else:
    print('all steps bit exact')
