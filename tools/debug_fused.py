import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_restoration_amd import ops
from oracle import restormer_ref as R
from oracle.fixtures import seeded_input
DEV="cuda"
def rel(a,b):
    b=b.double().cpu(); return float((a.double().cpu()-b).abs().max()/b.abs().max().clamp_min(1e-30))
def run(c, TH, PC, cfg):
    os.environ["MI_FG_CFG"]=cfg
    H,W=16,64
    sd = R.make_block_state(c, 1, 2.66, False, "WithBias", seed=31+c)
    y = seeded_input((1,c,H,W), 4100+c); yb=y.to(DEV).to(torch.bfloat16)
    ln_w=sd["norm2.body.weight"].to(DEV).float(); ln_b=sd["norm2.body.bias"].to(DEV).float()
    keys=["ffn.project_in.weight","ffn.project_in.bias","ffn.dwconv.weight","ffn.dwconv.bias","ffn.project_out.weight","ffn.project_out.bias"]
    params=tuple(sd[k].to(DEV).float().contiguous() if k in sd else None for k in keys)
    h=params[4].shape[1]
    pack=ops.gdfn_fused_pack(yb, ln_w, ln_b, params)
    HR=TH+2; BODY=HR*64; HPX=BODY+2*HR; HPXP=(HPX+15)//16*16
    # reference tensors (fp32 on cpu)
    yf=yb.float().cpu()
    xn=R.layernorm_nchw(yf, torch.ones(c), torch.zeros(c), "WithBias")[0]      # [C,H,W] (affine folded into W)
    w1=(sd["ffn.project_in.weight"].reshape(2*h,c)*sd["norm2.body.weight"][None,:])
    b1=sd["ffn.project_in.weight"].reshape(2*h,c)@sd["norm2.body.bias"]
    h0=torch.einsum("mk,khw->mhw", w1, xn)+b1[:,None,None]
    conv=F.conv2d(h0[None], sd["ffn.dwconv.weight"], None, padding=1, groups=2*h)[0]
    g=F.gelu(conv[:h])*conv[h:]
    # tile 0 = rows 0..TH-1, cols 0..63 ; halo rows -1..TH ; body linear px = r*64+col, Y=r-1
    for stage in (1,4,2,3):
        os.environ["MI_FG_DEBUG"]=str(stage)
        out,_,_=ops.gdfn_fused_fwd(yb, pack, h, True, want_stats=False)
        torch.cuda.synchronize()
        flat=out.float().cpu().reshape(-1)
        if stage==1:
            got=flat[:HPXP*c].reshape(HPXP,c)
            ref=torch.zeros(HPXP,c)
            for r in range(HR):
                Y=r-1
                if 0<=Y<H: ref[r*64:(r+1)*64]=xn[:,Y,:64].t()
            # halo columns: left X=-1 invalid (zeros in, LN of zeros = 0), right X=64 invalid since W=64
            print(f"  stage1 LN frags: body err {rel(got[:BODY], ref[:BODY]):.3e}  max|got| {float(got.abs().max()):.3f}")
            e=(got[:BODY]-ref[:BODY]).abs()
            print("    err by channel:", [round(float(v),3) for v in e.amax(dim=0)])
        elif stage==4:
            got=flat[:2*PC*(c+8)].reshape(2*PC,c+8)[:, :c]
            ref=torch.zeros(2*PC,c)
            for rr in range(2*PC):
                hid=(rr//PC)*h+rr%PC
                if rr%PC < h: ref[rr]=w1[hid]
            print(f"  stage4 W1 in LDS: err {rel(got, ref):.3e}")
        elif stage in (2,18):
            got=flat[:2*PC*HPXP].reshape(2*PC,HPXP)
            ref=torch.zeros(2*PC,HPXP)
            for rr in range(2*PC):
                hid=(rr//PC)*h+rr%PC
                for r in range(HR):
                    Y=r-1
                    if 0<=Y<H: ref[rr,r*64:(r+1)*64]=h0[hid,Y,:64]
            print(f"  stage{stage} h0 chunk0: err {rel(got, ref):.3e} max|got| {float(got.abs().max()):.3f} max|ref| {float(ref.abs().max()):.3f}")
            e=(got-ref).abs()
            print("    err by row(channel):", [round(float(v),2) for v in e.amax(dim=1)][:2*PC])
        else:
            got=flat[:PC*TH*64].reshape(PC,TH,64)
            ref=g[:PC,:TH,:64]
            print(f"  stage3 gate chunk0: err {rel(got, ref):.3e}")
    os.environ["MI_FG_DEBUG"]="0"
for c,TH,PC,cfg in [(96,8,32,"th8"),(96,8,16,"pc16"),(48,8,32,"th8,pc32"),(48,8,16,"th8,pc16")]:
    print(f"C={c} TH={TH} PC={PC}", flush=True)
    run(c,TH,PC,cfg)
