import sys; sys.path.insert(0,'.')
import numpy as np, torch
from pyimcom_amd import synth, smoke
from pyimcom_amd.stamps import PSFGroupTables, StampBatch
cfg=synth.CONFIGS['tiny']; stamps=[synth.make_stamp(cfg,i) for i in range(2)]
psfs,target=synth.make_psfs(cfg,max(s.n_expo for s in stamps))
tabs=PSFGroupTables(psfs,target,cfg.nfft); b=StampBatch(cfg,stamps,tabs); b.build(); torch.cuda.synchronize()
g,tref,Cref=smoke.oracle_tables(cfg,psfs,target); t_gpu=tabs.tables.cpu().numpy()
pt,pp,io=tabs.pair_maps(cfg.flat_penalty)
for k,st in enumerate(stamps):
    ref=smoke.oracle_stamp(cfg,g,t_gpu,tabs.C,st,pt,pp,io)
    n=st.n; A=b.A[k,:n,:n].cpu().numpy()
    d=np.abs(A-ref['A']); print("stamp",k,"n",n,"nan",np.isnan(A).sum(),"max err",np.nanmax(d),"max A",np.abs(ref['A']).max())
    bad=np.argwhere(d>1e-9)
    print(" nbad",len(bad), bad[:10].tolist())
    if len(bad):
        i,j=bad[0]; print(" A",A[i,j],"ref",ref['A'][i,j], "i%16",i%16,"j%16",j%16)
