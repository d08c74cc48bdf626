"""Token alignment between source and target prompt (host side, a few hundred integer ops per image).
The alignment conventions (word -> token counting rule, replacement matrix, refinement mapper) are those of prompt-to-prompt's
seq_aligner.py (github.com/google/prompt-to-prompt, Copyright 2022 Google LLC, Apache License 2.0), which the reference vendors as
modules/utils/seq_aligner.py; this file re-implements them and is checked against tables produced by that code (tests/golden/ptp_tables.npz).
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


def word_token_spans(text: str, tokenizer):
    """For every whitespace word of `text`: the positions (1-based, after BOS) of its tokens.  A word owns the following token pieces until
    their decoded characters cover its length -- the counting rule of prompt-to-prompt's get_word_inds."""
    words = text.split(" ")
    pieces = [tokenizer.decode([t]).strip("#") for t in tokenizer.encode(text)][1:-1]
    spans = [[] for _ in words]
    word = covered = 0
    for pos, piece in enumerate(pieces, start=1):
        spans[word].append(pos)
        covered += len(piece)
        if covered >= len(words[word]):
            word, covered = word + 1, 0
    return words, spans


def get_word_inds(text: str, word_place, tokenizer) -> np.ndarray:
    """token positions of a word given by string (every occurrence), by word index or by a list of word indices"""
    words = text.split(" ")
    if isinstance(word_place, str):
        wanted = [i for i, w in enumerate(words) if w == word_place]
    elif isinstance(word_place, int):
        wanted = [word_place]
    else:
        wanted = list(word_place)
    if not wanted:
        return np.array([], dtype=np.int64)
    _, spans = word_token_spans(text, tokenizer)
    return np.array(sorted(p for w in set(wanted) for p in spans[w]), dtype=np.int64)


def get_replacement_mapper_(x: str, y: str, tokenizer, max_len=77) -> torch.Tensor:
    """(max_len, max_len) matrix M of AttentionReplace: target probabilities = source probabilities @ M.  Tokens of unchanged words map
    one to one (in running source / target positions), a replaced word spreads its source tokens over the target word's tokens (1 each
    for equal token counts, else 1 / #target tokens), and everything after the last replaced word is the identity on the TARGET position
    (prompt-to-prompt's convention, including its stop when either running position reaches max_len)."""
    (wx, sx), (wy, sy) = word_token_spans(x, tokenizer), word_token_spans(y, tokenizer)
    if len(wx) != len(wy):
        raise ValueError(f"attention replacement edit can only be applied on prompts with the same length"
                         f" but prompt A has {len(wx)} words and prompt B has {len(wy)} words.")
    m = np.zeros((max_len, max_len))
    i = j = 0                                        # running source / target token position
    for w in range(len(wy)):
        if wy[w] == wx[w]:
            continue
        src, tgt = np.array(sx[w]), np.array(sy[w])
        same = int(src[0]) - i                       # unchanged tokens in front of this word
        if same > 0:
            m[np.arange(i, i + same), np.arange(j, j + same)] = 1
            i, j = i + same, j + same
        if len(src) == len(tgt):
            m[src, tgt] = 1
        else:
            m[np.ix_(src, tgt)] = 1 / len(tgt)
        i, j = i + len(src), j + len(tgt)
    tail = np.arange(j, j + max(0, max_len - max(i, j)))
    m[tail, tail] = 1
    return torch.from_numpy(m).float()


def get_replacement_mapper(prompts, tokenizer, max_len=77) -> torch.Tensor:
    return torch.stack([get_replacement_mapper_(prompts[0], p, tokenizer, max_len) for p in prompts[1:]])
