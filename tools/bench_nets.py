"""Time the native VAE (512x512 encode / decode) and CLIP text encoder; prints ms per call."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "eta-inversion_amd"))
from etainv.nets import NativeVAE, NativeCLIPText

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

dt = torch.bfloat16 if "bf16" in sys.argv else torch.float16
vae, clip = NativeVAE(None, dt), NativeCLIPText(None, dt)
for b in (1, 4):
    img = torch.rand(b, 3, 512, 512, device="cuda") * 2 - 1
    z = torch.randn(b, 4, 64, 64, device="cuda")
    ids = torch.randint(0, 49408, (b, 77), device="cuda")
    print(f"B={b}: vae.encode {timeit(lambda: vae.encode(img)):.2f} ms  vae.decode {timeit(lambda: vae.decode(z)):.2f} ms  "
          f"clip {timeit(lambda: clip(ids)):.2f} ms  (VAE FLOPs/img: enc 1.13 T, dec 2.54 T approx)")
