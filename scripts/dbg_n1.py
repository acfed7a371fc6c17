import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vlnce_amd
from vlnce_amd import data_path
from vlnce_amd.il_harness import update_agent
dev="cuda:0"
rng=np.random.RandomState(0)
T0=int(sys.argv[1]) if len(sys.argv)>1 else 100
lens=[max(1,int(T0*f)) for f in np.linspace(1.0,0.55,5)]
trajs=[]
for T in lens:
    obs={"rgb_features":rng.rand(T,2048,4,4).astype(np.float16),"depth_features":rng.rand(T,128,4,4).astype(np.float16),
         "instruction":np.tile(np.concatenate([rng.randint(1,2504,size=80),np.zeros(120,np.int64)])[None],(T,1))}
    oracle=rng.randint(0,4,size=T).astype(np.int64)
    trajs.append((obs,np.concatenate([[0],oracle[:-1]]).astype(np.int64),oracle))
out=data_path.collate_trajectories(trajs,dev,inflection_coef=3.2)
print("collated finite:", {k: bool(torch.isfinite(v).all()) for k,v in out[0].items()}, out[4].sum().item())
torch.manual_seed(0)
policy=vlnce_amd.build_model(vlnce_amd.make_config("CMAPolicy"),*vlnce_amd.make_spaces(256,256)).to(dev)
opt=torch.optim.Adam(policy.parameters(),lr=2.5e-4)
torch.distributions.Distribution.set_default_validate_args(False)
for i in range(4):
    loss,_,_=update_agent(policy,opt,*out[:5],512)
    bad=[n for n,p in policy.named_parameters() if not torch.isfinite(p).all()]
    print(i,"loss",float(loss),"nonfinite params:",bad[:4], flush=True)
