"""Prompt-to-prompt semantics restated for the CPU oracle (test infrastructure).

Follows (reference file:line):
  * token alignment (Needleman-Wunsch, gap 0 / match 1 / mismatch -1), refinement mapper
        modules/utils/seq_aligner.py:67-134
  * replacement mapper                         modules/utils/seq_aligner.py:158-201
  * word -> token indices                      modules/utils/ptp_utils.py:305-323
  * per-step cross-replace alpha table         modules/utils/ptp_utils.py:326-357
  * equalizer                                  modules/utils/ptp.py:277-286
  * AttentionControl / AttentionStore / AttentionControlEdit / Refine / Reweight / Replace
        modules/utils/ptp.py:91-274
  * LocalBlend                                 modules/utils/ptp.py:18-73
  * aggregate_attention / get_attention_map    modules/utils/ptp.py:288-303,
        modules/editing/ptp_editor.py:43-85
The reference hard-codes 64x64 latents / 16x16 maps / a 32^2 self-replace threshold
(SURVEY App. E-5); here they are parameters (`res`, `thres_n`) whose defaults reproduce it.
"""
import zlib
import numpy as np
import torch
import torch.nn.functional as F

MAX_NUM_WORDS = 77


class WordTokenizer:
    """Word-level stand-in for the CLIP BPE tokenizer (the real vocab is not in this image;
    SURVEY §8c item 4).  One token per whitespace word; BOS 49406 / EOS(=pad) 49407 like CLIP."""
    model_max_length = 77
    bos, eos = 49406, 49407

    def __init__(self):
        self._rev = {self.bos: "<|startoftext|>", self.eos: "<|endoftext|>"}

    def _id(self, w: str) -> int:
        i = 1000 + zlib.crc32(w.encode()) % 40000
        self._rev[i] = w
        return i

    def encode(self, text: str):
        return [self.bos] + [self._id(w) for w in text.split(" ") if w != ""] + [self.eos]

    def decode(self, ids):
        return " ".join(self._rev[int(i)] for i in ids)

    def pad_ids(self, text: str):
        ids = self.encode(text)[: self.model_max_length]
        return ids + [self.eos] * (self.model_max_length - len(ids))


# --------------------------------------------------------------------------- host-side tables
def global_align(x, y, gap=0, match=1, mismatch=-1):
    """seq_aligner.py:67-82 (ties prefer left, then up, then diag)."""
    nx, ny = len(x), len(y)
    m = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    m[0, 1:] = (np.arange(ny) + 1) * gap
    m[1:, 0] = (np.arange(nx) + 1) * gap
    tb = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    tb[0, 1:], tb[1:, 0], tb[0, 0] = 1, 2, 4
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            left = m[i, j - 1] + gap
            up = m[i - 1, j] + gap
            diag = m[i - 1, j - 1] + (match if x[i - 1] == y[j - 1] else mismatch)
            m[i, j] = max(left, up, diag)
            tb[i, j] = 1 if m[i, j] == left else 2 if m[i, j] == up else 3
    return m, tb


def aligned_mapper(x, y, tb):
    """seq_aligner.py:85-110: list of (j, i or -1) for every y token."""
    i, j, out = len(x), len(y), []
    while i > 0 or j > 0:
        if tb[i, j] == 3:
            i, j = i - 1, j - 1
            out.append((j, i))
        elif tb[i, j] == 1:
            j -= 1
            out.append((j, -1))
        elif tb[i, j] == 2:
            i -= 1
        else:
            break
    out.reverse()
    return np.array(out, dtype=np.int64).reshape(-1, 2)


def refinement_mapper(src: str, tgt: str, tok, max_len=MAX_NUM_WORDS):
    """seq_aligner.py:113-134 for one (src, tgt) pair -> mapper int64[77], alphas float32[77]."""
    x, y = tok.encode(src), tok.encode(tgt)
    _, tb = global_align(x, y)
    base = aligned_mapper(x, y, tb)
    alphas = np.ones(max_len, dtype=np.float32)
    alphas[: base.shape[0]] = (base[:, 1] != -1).astype(np.float32)
    mapper = np.zeros(max_len, dtype=np.int64)
    mapper[: base.shape[0]] = base[:, 1]
    mapper[base.shape[0]:] = len(y) + np.arange(max_len - len(y))
    return mapper, alphas


def word_inds(text: str, word_place, tok) -> np.ndarray:
    """ptp_utils.py:305-323."""
    split = text.split(" ")
    if isinstance(word_place, str):
        word_place = [i for i, w in enumerate(split) if w == word_place]
    elif isinstance(word_place, int):
        word_place = [word_place]
    out = []
    if len(word_place) > 0:
        enc = [tok.decode([t]).strip("#") for t in tok.encode(text)][1:-1]
        cur, ptr = 0, 0
        for i, piece in enumerate(enc):
            cur += len(piece)
            if ptr in word_place:
                out.append(i + 1)
            if cur >= len(split[ptr]):
                ptr += 1
                cur = 0
    return np.array(out, dtype=np.int64)


def replacement_mapper(src: str, tgt: str, tok, max_len=MAX_NUM_WORDS) -> np.ndarray:
    """seq_aligner.py:158-192 -> float32[77,77]."""
    wx, wy = src.split(" "), tgt.split(" ")
    if len(wx) != len(wy):
        raise ValueError("attention replacement edit can only be applied on prompts with the same length")
    rep = [i for i in range(len(wy)) if wy[i] != wx[i]]
    isrc = [word_inds(src, i, tok) for i in rep]
    itgt = [word_inds(tgt, i, tok) for i in rep]
    m = np.zeros((max_len, max_len))
    i = j = cur = 0
    while i < max_len and j < max_len:
        if cur < len(isrc) and isrc[cur][0] == i:
            s_, t_ = isrc[cur], itgt[cur]
            if len(s_) == len(t_):
                m[s_, t_] = 1
            else:
                for it in t_:
                    m[s_, it] = 1 / len(t_)
            cur += 1
            i += len(s_)
            j += len(t_)
        elif cur < len(isrc):
            m[i, j] = 1
            i += 1
            j += 1
        else:
            m[j, j] = 1
            i += 1
            j += 1
    return m.astype(np.float32)


def time_words_alpha(prompts, S: int, cross_replace_steps, tok, max_len=MAX_NUM_WORDS) -> np.ndarray:
    """ptp_utils.py:326-357 -> float32[S+1, n_prompts-1, 77]."""
    if not isinstance(cross_replace_steps, dict):
        cross_replace_steps = {"default_": cross_replace_steps}
    if "default_" not in cross_replace_steps:
        cross_replace_steps["default_"] = (0.0, 1.0)
    a = np.zeros((S + 1, len(prompts) - 1, max_len), dtype=np.float32)

    def upd(bounds, p, inds=None):
        if isinstance(bounds, float):
            bounds = (0, bounds)
        s, e = int(bounds[0] * a.shape[0]), int(bounds[1] * a.shape[0])
        inds_ = np.arange(max_len) if inds is None else inds
        a[:s, p, inds_] = 0
        a[s:e, p, inds_] = 1
        a[e:, p, inds_] = 0

    for i in range(len(prompts) - 1):
        upd(cross_replace_steps["default_"], i)
    for key, item in cross_replace_steps.items():
        if key != "default_":
            for i in range(1, len(prompts)):
                ind = word_inds(prompts[i], key, tok)
                if len(ind) > 0:
                    upd(item, i - 1, ind)
    return a


def equalizer(text: str, words, values, tok) -> np.ndarray:
    """ptp.py:277-286 -> float32[77]."""
    if isinstance(words, (int, str)):
        words = (words,)
    eq = np.ones(MAX_NUM_WORDS, dtype=np.float32)
    for w, v in zip(words, values):
        eq[word_inds(text, w, tok)] = v
    return eq


def blend_alpha_layers(prompts, words, tok) -> np.ndarray:
    """ptp.py:51-57 -> float32[n_prompts, 77]."""
    al = np.zeros((len(prompts), MAX_NUM_WORDS), dtype=np.float32)
    for i, (p, ws) in enumerate(zip(prompts, words)):
        if isinstance(ws, str):
            ws = [ws]
        for w in ws:
            al[i, word_inds(p, w, tok)] = 1
    return al


# --------------------------------------------------------------------------- controllers
def _empty_store():
    return {"down_cross": [], "mid_cross": [], "up_cross": [], "down_self": [], "mid_self": [], "up_self": []}


class AttentionStore:
    """ptp.py:91-183.  `__call__` sees the full (B*heads, N, M) probabilities of one attention
    layer and edits/stores only the cond half."""

    def __init__(self, num_att_layers=32, store_max_n=32 ** 2):
        self.num_att_layers = num_att_layers
        self.store_max_n = store_max_n
        self.cur_step = 0
        self.cur_att_layer = 0
        self.step_store = _empty_store()
        self.attention_store = {}

    def forward(self, attn, is_cross, place):
        if attn.shape[1] <= self.store_max_n:
            self.step_store[f"{place}_{'cross' if is_cross else 'self'}"].append(attn)
        return attn

    def between_steps(self):
        if len(self.attention_store) == 0:
            self.attention_store = self.step_store
        else:
            for key in self.attention_store:
                for i in range(len(self.attention_store[key])):
                    self.attention_store[key][i] += self.step_store[key][i]
        self.step_store = _empty_store()

    def count_layer(self):
        """Counter part of `__call__` alone (for layers the controller provably ignores)."""
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()

    def __call__(self, attn, is_cross, place):
        h = attn.shape[0]
        attn[h // 2:] = self.forward(attn[h // 2:], is_cross, place)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()
        return attn

    def step_callback(self, x_t):
        return x_t

    def average_attention(self):
        return {k: [it / self.cur_step for it in v] for k, v in self.attention_store.items()}


def aggregate_attention(store: AttentionStore, res: int, from_where, num_prompts: int, select: int, mid_res=8):
    """ptp.py:288-303 (cross maps only).  `mid_res` = side of the mid-block map (8 at 64x64 latents)."""
    out = []
    maps = store.average_attention()
    if res == mid_res:
        from_where = ["mid"]
    for loc in from_where:
        for item in maps[f"{loc}_cross"]:
            if item.shape[1] == res * res:
                out.append(item.reshape(num_prompts, -1, res, res, item.shape[-1])[select])
    out = torch.cat(out, dim=0)
    return out.sum(0) / out.shape[0]


def attention_map(store, token_idx: int, res=16, from_where=("up", "down"), resize=64, num_prompts=1, select=0):
    """ptp_editor.py:43-85: per-token map / max, bicubic -> (1,resize,resize), clamp [0,1]."""
    # (`res == 8` in the reference = the mid block's side at its 64 x 64 latents: resize / 8 here; a caller without `resize` keeps the default pairing)
    m = aggregate_attention(store, res, from_where, num_prompts, select, mid_res=resize // 8 if resize else res // 2)[:, :, token_idx][None]
    m = m / m.max()
    if resize is not None and m.shape[-2:] != (resize, resize):
        m = F.interpolate(m[None], (resize, resize), mode="bicubic")[0].clamp(0, 1)
    return m


class LocalBlend:
    """ptp.py:18-73 (no substruct words)."""

    def __init__(self, alpha_layers: np.ndarray, S: int, res=16, start_blend=0.2, th=0.3):
        self.alpha = torch.from_numpy(alpha_layers).reshape(alpha_layers.shape[0], 1, 1, 1, 1, MAX_NUM_WORDS)
        self.start_blend = int(start_blend * S)
        self.counter = 0
        self.th = th
        self.res = res

    def mask(self, x_t, attention_store):
        maps = attention_store["down_cross"][2:4] + attention_store["up_cross"][:3]
        maps = [it.reshape(self.alpha.shape[0], -1, 1, self.res, self.res, MAX_NUM_WORDS) for it in maps]
        maps = torch.cat(maps, dim=1)
        m = (maps * self.alpha.to(maps.dtype)).sum(-1).mean(1)
        m = F.max_pool2d(m, (3, 3), (1, 1), padding=(1, 1))
        m = F.interpolate(m, size=x_t.shape[2:])
        m = m / m.max(2, keepdim=True)[0].max(3, keepdim=True)[0]
        m = m.gt(self.th)
        return m[:1] + m

    def __call__(self, x_t, attention_store):
        self.counter += 1
        if self.counter > self.start_blend:
            m = self.mask(x_t, attention_store).to(x_t.dtype)
            x_t = x_t[:1] + m * (x_t - x_t[:1])
        return x_t


class AttentionEdit(AttentionStore):
    """AttentionControlEdit + {Refine | Replace} (+ Reweight) for exactly two prompts
    (ptp.py:186-274, make_controller ptp.py:306-320)."""

    def __init__(self, S, cross_alpha, self_replace_steps, mapper=None, alphas=None, replace_matrix=None,
                 equalizer=None, local_blend=None, thres_n=32 ** 2, **kw):
        super().__init__(**kw)
        self.batch_size = 2
        self.cross_alpha = torch.from_numpy(cross_alpha)            # (S+1, 1, 77)
        if isinstance(self_replace_steps, float):
            self_replace_steps = (0, self_replace_steps)
        self.num_self_replace = (int(S * self_replace_steps[0]), int(S * self_replace_steps[1]))
        self.mapper = None if mapper is None else torch.from_numpy(mapper)
        self.alphas = None if alphas is None else torch.from_numpy(alphas)
        self.replace_matrix = None if replace_matrix is None else torch.from_numpy(replace_matrix)
        self.equalizer = None if equalizer is None else torch.from_numpy(equalizer)
        self.local_blend = local_blend
        self.thres_n = thres_n

    def replace_cross(self, base, repl):
        # base (h,N,77), repl (1,h,N,77)
        if self.replace_matrix is not None:                       # AttentionReplace ptp.py:236-237
            out = torch.einsum("hpw,wn->hpn", base, self.replace_matrix.to(base.dtype))[None]
        else:                                                     # AttentionRefine ptp.py:247-251
            a = self.alphas.to(base.dtype)
            out = base[:, :, self.mapper][None] * a + repl * (1 - a)
        if self.equalizer is not None:                            # AttentionReweight ptp.py:263-268
            out = out * self.equalizer.to(base.dtype)
        return out

    def forward(self, attn, is_cross, place):
        super().forward(attn, is_cross, place)
        if is_cross or (self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]):
            h = attn.shape[0] // self.batch_size
            attn = attn.reshape(self.batch_size, h, *attn.shape[1:])
            base, repl = attn[0], attn[1:]
            if is_cross:
                aw = self.cross_alpha[self.cur_step].to(attn.dtype)          # (1,77)
                attn[1:] = self.replace_cross(base, repl) * aw + (1 - aw) * repl
            elif repl.shape[2] <= self.thres_n:
                attn[1:] = base.unsqueeze(0).expand(repl.shape[0], *base.shape)
            attn = attn.reshape(self.batch_size * h, *attn.shape[2:])
        return attn

    def step_callback(self, x_t):
        if self.local_blend is not None:
            x_t = self.local_blend(x_t, self.attention_store)
        return x_t


def make_edit_controller(src, tgt, S, tok, is_replace_controller=False, cross_replace_steps=None,
                         self_replace_steps=0.6, blend_words=None, equilizer_params=None, res=16,
                         thres_n=32 ** 2, **unused) -> AttentionEdit:
    """ptp.make_controller (ptp.py:306-320) for prompts=[src, tgt]."""
    prompts = [src, tgt]
    ca = time_words_alpha(prompts, S, cross_replace_steps if cross_replace_steps is not None else {"default_": 0.8}, tok)
    lb = None
    if blend_words is not None:
        lb = LocalBlend(blend_alpha_layers(prompts, blend_words, tok), S, res=res)
    kw = dict(local_blend=lb, thres_n=thres_n, store_max_n=thres_n)
    if equilizer_params is not None:
        kw["equalizer"] = equalizer(tgt, equilizer_params["words"], equilizer_params["values"], tok)
    if is_replace_controller:
        kw["replace_matrix"] = replacement_mapper(src, tgt, tok)
    else:
        kw["mapper"], kw["alphas"] = refinement_mapper(src, tgt, tok)
    return AttentionEdit(S, ca, self_replace_steps, **kw)
