"""Token alignment between source and target prompt (host side, a few hundred integer ops per image).
Interface of the reference's modules/utils/seq_aligner.py: get_refinement_mapper (:127-134), get_mapper (:113-124),
get_replacement_mapper (:195-201), get_word_inds (:137-155).  The global alignment is Needleman-Wunsch with
gap 0 / match +1 / mismatch -1 and the reference's tie order (left, then up, then diagonal; :67-82)."""
import numpy as np
import torch


def _align(x, y):
    """traceback codes: 1 = gap in x (consume y), 2 = gap in y (consume x), 3 = pair, 4 = origin"""
    nx, ny = len(x), len(y)
    score = np.zeros((nx + 1, ny + 1), dtype=np.int64)     # gap penalty 0 -> borders stay 0
    back = np.zeros((nx + 1, ny + 1), dtype=np.int8)
    back[0, 1:], back[1:, 0], back[0, 0] = 1, 2, 4
    for i in range(1, nx + 1):
        xi = x[i - 1]
        for j in range(1, ny + 1):
            left, up = score[i, j - 1], score[i - 1, j]
            diag = score[i - 1, j - 1] + (1 if xi == y[j - 1] else -1)
            best = max(left, up, diag)
            score[i, j] = best
            back[i, j] = 1 if best == left else (2 if best == up else 3)
    return back


def _y_to_x(x, y, back):
    i, j, pairs = len(x), len(y), []
    while i > 0 or j > 0:
        code = back[i, j]
        if code == 3:
            i, j = i - 1, j - 1
            pairs.append((j, i))
        elif code == 1:
            j -= 1
            pairs.append((j, -1))
        elif code == 2:
            i -= 1
        else:
            break
    return pairs[::-1]


def get_mapper(x: str, y: str, tokenizer, max_len=77):
    xs, ys = tokenizer.encode(x), tokenizer.encode(y)
    pairs = _y_to_x(xs, ys, _align(xs, ys))
    src = torch.tensor([p[1] for p in pairs], dtype=torch.int64)
    alphas = torch.ones(max_len)
    alphas[: len(pairs)] = (src != -1).float()
    mapper = torch.zeros(max_len, dtype=torch.int64)
    mapper[: len(pairs)] = src
    mapper[len(pairs):] = len(ys) + torch.arange(max_len - len(ys))
    return mapper, alphas


def get_refinement_mapper(prompts, tokenizer, max_len=77):
    out = [get_mapper(prompts[0], p, tokenizer, max_len) for p in prompts[1:]]
    return torch.stack([m for m, _ in out]), torch.stack([a for _, a in out])


def get_word_inds(text: str, word_place, tokenizer) -> np.ndarray:
    """token positions (1-based, after BOS) of a word given by string or by word index"""
    words = text.split(" ")
    if isinstance(word_place, str):
        wanted = {i for i, w in enumerate(words) if w == word_place}
    elif isinstance(word_place, int):
        wanted = {word_place}
    else:
        wanted = set(word_place)
    out = []
    if wanted:
        pieces = [tokenizer.decode([t]).strip("#") for t in tokenizer.encode(text)][1:-1]
        word, consumed = 0, 0
        for pos, piece in enumerate(pieces):
            consumed += len(piece)
            if word in wanted:
                out.append(pos + 1)
            if consumed >= len(words[word]):
                word, consumed = word + 1, 0
    return np.array(out, dtype=np.int64)


def get_replacement_mapper_(x: str, y: str, tokenizer, max_len=77) -> torch.Tensor:
    wx, wy = x.split(" "), y.split(" ")
    if len(wx) != len(wy):
        raise ValueError(f"attention replacement edit can only be applied on prompts with the same length"
                         f" but prompt A has {len(wx)} words and prompt B has {len(wy)} words.")
    changed = [i for i in range(len(wy)) if wy[i] != wx[i]]
    src = [get_word_inds(x, i, tokenizer) for i in changed]
    tgt = [get_word_inds(y, i, tokenizer) for i in changed]
    m = np.zeros((max_len, max_len))
    i = j = k = 0
    while i < max_len and j < max_len:
        if k < len(src) and src[k][0] == i:
            if len(src[k]) == len(tgt[k]):
                m[src[k], tgt[k]] = 1
            else:
                for jt in tgt[k]:
                    m[src[k], jt] = 1 / len(tgt[k])
            i, j, k = i + len(src[k]), j + len(tgt[k]), k + 1
        elif k < len(src):
            m[i, j] = 1
            i, j = i + 1, j + 1
        else:
            m[j, j] = 1
            i, j = i + 1, j + 1
    return torch.from_numpy(m).float()


def get_replacement_mapper(prompts, tokenizer, max_len=77) -> torch.Tensor:
    return torch.stack([get_replacement_mapper_(prompts[0], p, tokenizer, max_len) for p in prompts[1:]])
