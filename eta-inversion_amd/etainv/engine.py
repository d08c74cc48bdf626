"""Python handle on the native UNet executor (csrc/engine.cpp) -- plumbing only: it owns nothing but the opaque
engine pointer and passes device pointers of torch tensors through the C ABI."""
import contextlib
import ctypes as C
import os

import torch

from . import _capi
from .weights import load_snapshot, synthetic_tensor


class AttnControl:
    """Declarative description of what the reference's attention hooks do during one UNet call
    (reference modules/utils/ptp_utils.py:196-302, modules/utils/masactrl_utils.py:74-153)."""

    def __init__(self, mode=_capi.ATTN_PLAIN, n_img=1, store_maps=False, mapper=None, alphas=None, replace_mat=None,
                 equalizer=None, cross_alpha=None, self_replace_active=False, self_max_tokens=32 ** 2, masa_active=False,
                 masa_first_block=10, first_row=0, src_exit_block=0):
        self._keep = (mapper, alphas, replace_mat, equalizer, cross_alpha)
        self.c = _capi.AttnCtrl(mode=mode, n_img=n_img, store_maps=int(store_maps), mapper=_capi.ptr(mapper),
                                alphas=_capi.ptr(alphas), replace_mat=_capi.ptr(replace_mat), equalizer=_capi.ptr(equalizer),
                                cross_alpha=_capi.ptr(cross_alpha), self_replace_active=int(self_replace_active),
                                self_max_tokens=int(self_max_tokens), masa_active=int(masa_active),
                                masa_first_block=int(masa_first_block), first_row=int(first_row), src_exit_block=int(src_exit_block))


class Engine:
    _ctx_generation = 0

    def __init__(self, dtype=torch.float16, max_unet_batch=4, latent_size=64, max_img=1, device="cuda:0"):
        if not torch.cuda.is_available():
            raise _capi.EtainvError("no HIP device: the etainv engine has no CPU fallback")
        self.lib = _capi.load()
        self.dtype, self.L, self.max_unet_batch, self.max_img = dtype, latent_size, max_unet_batch, max_img
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        cfg = _capi.EngineConfig(compute_dtype=_capi.dtype_code(dtype), max_unet_batch=max_unet_batch, latent_size=latent_size,
                                 max_img=max_img)
        h = C.c_void_p()
        _capi.check(self.lib.etainv_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.map_div = 4

    def close(self):
        if getattr(self, "h", None):
            self.lib.etainv_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def weight_specs(self):
        n = self.lib.etainv_engine_num_weights(self.h)
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(n):
            _capi.check(self.lib.etainv_engine_weight_info(self.h, i, name, 256, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def set_weight(self, name, tensor):
        t = tensor.detach().to(device=self.device, dtype=torch.float32).contiguous()
        _capi.check(self.lib.etainv_engine_set_weight(self.h, name.encode(), _capi.ptr(t), t.numel(), _capi.stream_ptr()))
        torch.cuda.current_stream().synchronize()  # `t` may be freed after return

    def load_state_dict(self, sd):
        for name, shape in self.weight_specs():
            if name not in sd:
                raise KeyError(f"UNet parameter {name} missing from the state dict")
            if tuple(sd[name].shape) != shape:
                raise ValueError(f"{name}: expected {shape}, got {tuple(sd[name].shape)}")
            self.set_weight(name, sd[name])
        assert self.lib.etainv_engine_weights_ready(self.h) == 1

    def load_synthetic(self, seed=0):
        for name, shape in self.weight_specs():
            self.set_weight(name, synthetic_tensor(name, shape, seed))
        assert self.lib.etainv_engine_weights_ready(self.h) == 1

    def load_default(self, seed=0):
        """`ETAINV_SD_PATH` -> local diffusers snapshot, else deterministic synthetic weights."""
        path = os.environ.get("ETAINV_SD_PATH")
        if path:
            self.load_state_dict(load_snapshot(path))
        else:
            self.load_synthetic(seed)

    # ------------------------------------------------------------------ compute
    def unet(self, latent, t, ctx, ctrl=None, out=None):
        """eps = UNet(latent, t, ctx): latent (n_lat,4,L,L), ctx (rows,77,768); UNet row r uses latent r % n_lat."""
        rows, n_lat = ctx.shape[0], latent.shape[0]
        assert latent.dtype == ctx.dtype and latent.shape[1:] == (4, self.L, self.L) and ctx.shape[1:] == (77, 768)
        latent, ctx = latent.contiguous(), ctx.contiguous()
        if out is None:
            out = torch.empty(rows, 4, self.L, self.L, dtype=latent.dtype, device=latent.device)
        if isinstance(t, torch.Tensor):
            t = t.reshape(-1).tolist() if t.dim() else [int(t)]
        elif isinstance(t, (int, float)):
            t = [int(t)]
        t = list(t)
        if len(t) == 1:
            t = t * rows
        t_arr = (C.c_int64 * rows)(*[int(v) for v in t])
        _capi.check(self.lib.etainv_unet_forward(self.h, _capi.ptr(latent), n_lat, t_arr, _capi.ptr(ctx), rows,
                                                 C.byref(ctrl.c) if ctrl is not None else None, _capi.ptr(out),
                                                 _capi.dtype_code(latent.dtype), _capi.stream_ptr()))
        return out

    def maps_reset(self):
        _capi.check(self.lib.etainv_maps_reset(self.h, _capi.stream_ptr()))

    def word_maps(self, n_img, tokens, steps_done, out, accumulate=False, scale=1.0):
        _capi.check(self.lib.etainv_maps_word_maps(self.h, n_img, _capi.ptr(tokens), tokens.shape[1], steps_done, _capi.ptr(out),
                                                   int(accumulate), float(scale), _capi.stream_ptr()))
        return out

    def word_maps_role(self, n_img, tokens, steps_done, row_sel, out):
        """maps of one role of the backward-pass store (0 source, 1 target cond row), averaged over `steps_done` steps"""
        _capi.check(self.lib.etainv_maps_word_maps_role(self.h, n_img, _capi.ptr(tokens), tokens.shape[1], steps_done, row_sel, _capi.ptr(out),
                                                        0, 1.0, _capi.stream_ptr()))
        return out

    def maps_configure(self, res_div=4):
        """which cross layers the attention-map store keeps: 4 = the five (L/4)^2 layers (default), 2 = the (L/2)^2 layers, 8 = the mid block's; clears the store"""
        if int(res_div) not in (2, 4, 8) or self.L % int(res_div):
            raise ValueError(f"res_div must be 2, 4 or 8 and divide L = {self.L}, got {res_div}")
        self.map_div = 4                              # (what the library falls back to when its allocation fails)
        _capi.check(self.lib.etainv_maps_configure(self.h, int(res_div), _capi.stream_ptr()))
        self.map_div = int(res_div)

    def word_maps_ex(self, n_img, tokens, steps_done, row_sel, layer_mask, out, accumulate=False, scale=1.0):
        """word maps over the layers selected by `layer_mask` (bits: down 0x03, up 0x1c; res_div 8: mid 0x01)"""
        _capi.check(self.lib.etainv_maps_word_maps_ex(self.h, n_img, _capi.ptr(tokens), tokens.shape[1], steps_done, int(row_sel), int(layer_mask),
                                                      _capi.ptr(out), int(accumulate), float(scale), _capi.stream_ptr()))
        return out

    def cache_context(self, enable):
        """loops that pass one unchanged context tensor to every UNet call: reuse its cross-attention K / V projections (off when the loop ends)"""
        _capi.check(self.lib.etainv_engine_cache_context(self.h, int(bool(enable))))

    @contextlib.contextmanager
    def cached_context(self):
        """`with engine.cached_context():` around a loop whose UNet calls all pass the same, unchanged context tensor.  A fresh generation per
        loop (an allocator may hand a new tensor the old address) and the cache is switched off on every exit path, exceptions included."""
        Engine._ctx_generation += 1
        _capi.check(self.lib.etainv_engine_context_generation(self.h, Engine._ctx_generation))
        self.cache_context(True)
        try:
            yield
        finally:
            self.cache_context(False)

    def local_blend(self, x, n_img, blend_alpha, thres=0.3):
        assert x.dtype == torch.float32
        _capi.check(self.lib.etainv_local_blend(self.h, _capi.ptr(x), n_img, _capi.ptr(blend_alpha), float(thres), _capi.stream_ptr()))
        return x

    @property
    def graph_stats(self):
        """(captures, replays) of the hipGraph path of small UNet calls since the engine was created"""
        cap, rep = C.c_int64(), C.c_int64()
        _capi.check(self.lib.etainv_engine_graph_stats(self.h, C.byref(cap), C.byref(rep)))
        return cap.value, rep.value

    @property
    def qkv_head_major_launches(self):
        """fused QKV projections that wrote the head-major layout (ETAINV_QKV_HM=1 at engine creation) since the engine was created"""
        n = C.c_longlong()
        _capi.check(self.lib.etainv_engine_qkv_head_major_count(self.h, C.byref(n)))
        return n.value

    @property
    def workspace_bytes(self):
        return self.lib.etainv_engine_workspace_bytes(self.h)

    @property
    def weight_bytes(self):
        return self.lib.etainv_engine_weight_bytes(self.h)
