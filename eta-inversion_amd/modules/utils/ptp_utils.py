"""Host-side prompt-to-prompt tables (reference modules/utils/ptp_utils.py:305-357)."""
import torch

from .seq_aligner import get_word_inds  # noqa: F401  (same function, re-exported like the reference)


def update_alpha_time_word(alpha, bounds, prompt_ind, word_inds=None):
    if isinstance(bounds, float):
        bounds = (0, bounds)
    n = alpha.shape[0]
    start, end = int(bounds[0] * n), int(bounds[1] * n)
    if word_inds is None:
        word_inds = torch.arange(alpha.shape[2])
    alpha[:start, prompt_ind, word_inds] = 0
    alpha[start:end, prompt_ind, word_inds] = 1
    alpha[end:, prompt_ind, word_inds] = 0
    return alpha


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=77):
    """(num_steps+1, n_prompts-1, 1, 1, 77): 1 while the cross-attention edit is active for that token."""
    if not isinstance(cross_replace_steps, dict):
        cross_replace_steps = {"default_": cross_replace_steps}
    if "default_" not in cross_replace_steps:
        cross_replace_steps["default_"] = (0., 1.)
    alpha = torch.zeros(num_steps + 1, len(prompts) - 1, max_num_words)
    for i in range(len(prompts) - 1):
        alpha = update_alpha_time_word(alpha, cross_replace_steps["default_"], i)
    for word, bounds in cross_replace_steps.items():
        if word == "default_":
            continue
        for i in range(1, len(prompts)):
            inds = get_word_inds(prompts[i], word, tokenizer)
            if len(inds) > 0:
                alpha = update_alpha_time_word(alpha, bounds, i - 1, torch.as_tensor(inds))
    return alpha.reshape(num_steps + 1, len(prompts) - 1, 1, 1, max_num_words)
