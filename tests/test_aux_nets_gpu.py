"""VAE encode/decode and CLIP text encoder on the native kernels vs the CPU oracle (oracle/vae.py, oracle/clip.py), same
seeded synthetic weights by name.  Tolerance: fp16 operands / fp32 accumulation, relative L2 <= 5e-3 (bf16: 3e-2)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def vae_pair():
    from oracle.vae import build_vae
    from etainv.nets import NativeVAE
    return build_vae(0), NativeVAE(None, torch.float16, 0)


@pytest.mark.parametrize("size", [64, 128])
def test_vae_encode(vae_pair, size):
    ref, nat = vae_pair
    g = torch.Generator().manual_seed(size)
    img = torch.rand(2, 3, size, size, generator=g) * 2 - 1
    with torch.no_grad():
        want = ref.encode_mean(img)
    got = nat.encode(img.cuda())["latent_dist"].mean
    assert got.shape == want.shape and got.dtype == torch.float32
    assert rel(got, want) < 5e-3


@pytest.mark.parametrize("size", [8, 16])
def test_vae_decode(vae_pair, size):
    ref, nat = vae_pair
    g = torch.Generator().manual_seed(size)
    z = torch.randn(2, 4, size, size, generator=g)
    with torch.no_grad():
        want = ref.decode(z)
    got = nat.decode(z.cuda())["sample"]
    assert got.shape == want.shape and got.dtype == torch.float32
    assert rel(got, want) < 5e-3


def test_vae_bf16_and_state_dict():
    """weights handed over as a diffusers-named state dict (the ETAINV_SD_PATH route), bf16 operands"""
    from oracle.vae import build_vae
    from etainv.nets import NativeVAE
    ref = build_vae(3)
    nat = NativeVAE({k: v for k, v in ref.state_dict().items()}, torch.bfloat16)
    z = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = ref.decode(z)
    assert rel(nat.decode(z.cuda())["sample"], want) < 3e-2


def test_clip_text_encoder():
    from oracle.clip import build_clip
    from etainv.nets import NativeCLIPText
    ref, nat = build_clip(0), NativeCLIPText(None, torch.float16, 0)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 49408, (3, 77), generator=g)
    ids[:, 0], ids[:, 20:] = 49406, 49407
    with torch.no_grad():
        want = ref(ids)[0]
    got = nat(ids.cuda())[0]
    assert got.shape == (3, 77, 768) and got.dtype == torch.float32
    assert rel(got, want) < 5e-3
    # causal: changing a later token leaves earlier positions untouched (bit-exact)
    ids2 = ids.clone()
    ids2[:, 10] = 1234
    got2 = nat(ids2.cuda())[0]
    assert torch.equal(got2[:, :10], got[:, :10]) and not torch.equal(got2[:, 10:], got[:, 10:])
