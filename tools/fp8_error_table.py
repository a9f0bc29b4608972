import sys, torch
sys.path.insert(0, '/root/repo')
from druglamp_amd import ops
DEV='cuda:0'
def ref(q,k,v,scale):
    s=(q.double()@k.double().transpose(-1,-2))*scale
    return torch.softmax(s,-1)@v.double(), torch.logsumexp(s,-1)
for std, outl in ((0.5,1),(1.0,1),(1.0,4),(1.3,6)):
  for (P,H,S,Lq,Lk,hd) in [(4,4,2,1024,1024,64),(2,4,1,1024,1024,128),(3,2,1,300,333,64)]:
    d=H*hd
    g=torch.Generator().manual_seed(1)
    qkv=(torch.randn(P,max(Lq,Lk),3*d,generator=g)*std).to(torch.bfloat16).to(DEV)
    qkv[...,5]*=outl
    q,k,v=qkv[:,:Lq,:d],qkv[:,:Lk,d:2*d],qkv[:,:Lk,2*d:]
    Lm=max(Lq,Lk); st=(Lm*3*d,hd,3*d); shift=P//2 if S==2 else 0
    split=lambda t,L_: t.reshape(P,L_,H,hd).permute(0,2,1,3)
    ro,rl=ref(split(q,Lq),split(k,Lk),split(v,Lk),hd**-0.5)
    line="std %.1f outl %d %s:"%(std,outl,(P,H,S,Lq,Lk,hd))
    for fp8 in (False,True):
        o=torch.zeros(S,P,Lq,d,device=DEV,dtype=torch.bfloat16)
        lse=ops.attn_fwd(q,k,v,n_problems=P,n_heads=H,n_segments=S,partner_shift=shift,Lq=Lq,Lk=Lk,head_dim=hd,scale=hd**-0.5,q_strides=st,k_strides=st,v_strides=st,out=o,o_strides=(Lq*d,hd,d),o_ss=P*Lq*d,fp8=fp8)
        got=split(o[0],Lq).double()
        err=(got-ro).abs(); sc=float(ro.abs().max())
        line+="  %s max %.4f mean %.5f rms %.5f lse %.4f"%("fp8" if fp8 else "bf16",float(err.max())/sc,float(err.mean())/sc,float((err**2).mean().sqrt())/sc,float((lse[0].double()-rl).abs().max()))
    print(line,flush=True)
