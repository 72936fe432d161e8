import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from daisyrec_amd import ops
dev='cuda'
rng=np.random.default_rng(0)
U,I,d,B=50,40,32,64
P=torch.from_numpy((rng.standard_normal((U,d))*0.1).astype(np.float32)).to(dev)
Q=torch.from_numpy((rng.standard_normal((I,d))*0.1).astype(np.float32)).to(dev)
u,i,j=(torch.from_numpy(rng.integers(0,n,B).astype(np.int32)).to(dev) for n in (U,I,I))
ctx=ops.BprContext(B,d,U,I)
print('ctx ok'); 
ctx.set_batch(u,i,j); torch.cuda.synchronize(); print('set_batch ok')
ctx.forward(P,Q); torch.cuda.synchronize(); print('fwd ok', ctx.stats.cpu().numpy()[:7])
ctx.finalize(1e-3,1e-3); torch.cuda.synchronize(); print('fin ok', ctx.stats.cpu().numpy()[7:11])
for mode in (1,2,0):
    ctx.item_grad(P,Q,1e-3,1e-3,mode); torch.cuda.synchronize(); print('item ok', mode, float(ctx.gQ.abs().sum()))
    ctx.item_sgd_apply(Q,0.0); torch.cuda.synchronize(); print('apply ok', float(ctx.gQ.abs().sum()))
ctx.user_sgd(P,Q,0.01,1e-3,1e-3); torch.cuda.synchronize(); print('user ok')
sl=torch.zeros(1,dtype=torch.float64,device=dev)
ctx.sgd_step(P,Q,0.01,1e-3,1e-3,item_mode=2,step_loss=sl); torch.cuda.synchronize(); print('step ok', float(sl))
