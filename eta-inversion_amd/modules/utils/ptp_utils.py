"""Host-side prompt-to-prompt tables (reference modules/utils/ptp_utils.py:305-357)."""
import torch

from .seq_aligner import get_word_inds  # noqa: F401  (same function, re-exported like the reference)


def update_alpha_time_word(alpha, bounds, prompt_ind, word_inds=None):
    """alpha[step, prompt_ind, words] := 1 for steps in [lo * n, hi * n) (n = number of table rows), 0 outside; a single float bounds the window
    from 0 (reference :326-336)"""
    lo, hi = (0.0, bounds) if isinstance(bounds, float) else (bounds[0], bounds[1])
    rows = alpha.shape[0]
    window = torch.zeros(rows, dtype=alpha.dtype)
    window[int(lo * rows):int(hi * rows)] = 1
    cols = torch.arange(alpha.shape[2]) if word_inds is None else torch.as_tensor(word_inds)   # (get_word_inds returns a numpy array, reference :326)
    alpha[:, prompt_ind, cols] = window[:, None].expand(rows, len(cols)) if cols.dim() else window
    return alpha


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=77):
    """(num_steps+1, n_prompts-1, 1, 1, 77): 1 while the cross-attention edit is active for that token: the "default_" window for every token of
    every edited prompt, then per-word windows for the tokens of the words named in `cross_replace_steps` (reference :339-357)"""
    windows = dict(cross_replace_steps) if isinstance(cross_replace_steps, dict) else {"default_": cross_replace_steps}
    windows.setdefault("default_", (0., 1.))
    if isinstance(cross_replace_steps, dict):
        cross_replace_steps.setdefault("default_", (0., 1.))      # (the reference fills the caller's dict in)
    n_edit = len(prompts) - 1
    alpha = torch.zeros(num_steps + 1, n_edit, max_num_words)
    for e in range(n_edit):
        update_alpha_time_word(alpha, windows["default_"], e)
    for word, bounds in windows.items():
        if word == "default_":
            continue
        for e in range(n_edit):
            inds = get_word_inds(prompts[e + 1], word, tokenizer)
            if len(inds) > 0:
                update_alpha_time_word(alpha, bounds, e, torch.as_tensor(inds))
    return alpha.reshape(num_steps + 1, n_edit, 1, 1, max_num_words)
