import sys, torch
sys.path.insert(0, "eta-inversion_amd")
from etainv.engine import Engine
torch.manual_seed(0)
for dt in (torch.float16,):
    e = Engine(dtype=dt, max_unet_batch=16, latent_size=64, max_img=4)
    e.load_synthetic(0)
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(1, 4, 64, 64, generator=g).cuda(); c1 = torch.randn(1, 77, 768, generator=g).cuda()
    outs = {}
    for rows in (1, 2, 4, 8, 16):
        x = x1.repeat(rows, 1, 1, 1).contiguous(); c = c1.repeat(rows, 1, 1).contiguous()
        out = torch.empty(rows, 4, 64, 64, device="cuda")
        e.unet(x, 500, c, None, out=out)
        torch.cuda.synchronize()
        outs[rows] = out.clone()
        same = all(torch.equal(out[0], out[i]) for i in range(rows))
        print(rows, "rows identical within batch:", same, " vs rows=1 rel:", ((out[0] - outs[1][0]).norm() / outs[1][0].norm()).item(), "finite", bool(torch.isfinite(out).all()))
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle.unet import build_unet
torch.set_num_threads(32)
u = build_unet(0)
with torch.no_grad():
    ref = u(x1.cpu(), 500, encoder_hidden_states=c1.cpu())["sample"]
for rows in outs:
    print("rows", rows, "vs oracle rel:", ((outs[rows][0].cpu() - ref[0]).norm() / ref[0].norm()).item())
