"""Prompt-to-prompt controllers as DECLARATIVE descriptors.

The reference's controllers (modules/utils/ptp.py:91-320) are Python callbacks invoked once per attention layer on
materialised N x N probabilities.  Here a controller only holds the per-image tables; the edits themselves run inside
the HIP attention kernels (csrc/attention.hip) and LocalBlend in csrc/maps.hip.  Same constructor surface:
make_controller(model, prompts, is_replace_controller, cross_replace_steps, self_replace_steps, blend_words,
equilizer_params)."""
import numpy as np
import torch

from . import ptp_utils, seq_aligner

MAX_NUM_WORDS = 77


def get_equalizer(model, text, word_select, values):
    if isinstance(word_select, (int, str)):
        word_select = (word_select,)
    eq = torch.ones(1, MAX_NUM_WORDS)
    for word, val in zip(word_select, values):
        eq[:, ptp_utils.get_word_inds(text, word, model.tokenizer)] = val
    return eq


class LocalBlend:
    """tokens of the blend words per prompt + thresholds (reference ptp.py:49-73)"""

    def __init__(self, model, prompts, words, start_blend=0.2, th=(.3, .3)):
        alpha = torch.zeros(len(prompts), MAX_NUM_WORDS)
        for i, (prompt, ws) in enumerate(zip(prompts, words)):
            for w in ([ws] if isinstance(ws, str) else ws):
                alpha[i, ptp_utils.get_word_inds(prompt, w, model.tokenizer)] = 1
        self.alpha_layers = alpha
        self.start_blend = int(start_blend * model.scheduler.num_inference_steps)
        self.th = th


class AttentionControlEdit:
    """Tables of one (source, target) edit: Refine or Replace mapper, optional Reweight equalizer, LocalBlend."""

    def __init__(self, model, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend=None, is_replace=False,
                 equalizer=None):
        assert len(prompts) == 2, "exactly one (source, target) pair per controller (reference ptp.py:189)"
        tok = model.tokenizer
        self.prompts, self.num_steps = prompts, num_steps
        self.cross_replace_alpha = ptp_utils.get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tok)
        if isinstance(self_replace_steps, float):
            self_replace_steps = (0, self_replace_steps)
        self.self_replace_steps = self_replace_steps
        self.num_self_replace = (int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1]))
        self.local_blend = local_blend
        self.mapper = self.alphas = self.replace_matrix = None
        if is_replace:
            self.replace_matrix = seq_aligner.get_replacement_mapper(prompts, tok)[0]
        else:
            mapper, alphas = seq_aligner.get_refinement_mapper(prompts, tok)
            self.mapper, self.alphas = mapper[0], alphas[0]
        self.equalizer = None if equalizer is None else equalizer[0]

    def tables(self):
        """numpy tables in the layout etainv.pipeline.PtpTables stacks over images"""
        S = self.num_steps
        d = {"cross_alpha": self.cross_replace_alpha.reshape(S + 1, MAX_NUM_WORDS).numpy(),
             "mapper": None if self.mapper is None else self.mapper.numpy().astype(np.int32),
             "alphas": None if self.alphas is None else self.alphas.numpy(),
             "replace_mat": None if self.replace_matrix is None else self.replace_matrix.numpy(),
             "equalizer": None if self.equalizer is None else self.equalizer.numpy(),
             "blend_alpha": None if self.local_blend is None else self.local_blend.alpha_layers.numpy()}
        return d


def make_controller(model, prompts, is_replace_controller, cross_replace_steps, self_replace_steps, blend_words=None,
                    equilizer_params=None, **kwargs):
    S = model.scheduler.num_inference_steps
    lb = None if blend_words is None else LocalBlend(model, prompts, blend_words)
    eq = None
    if equilizer_params is not None:
        eq = get_equalizer(model, prompts[1], equilizer_params["words"], equilizer_params["values"])
    return AttentionControlEdit(model, prompts, S, cross_replace_steps, self_replace_steps, local_blend=lb,
                                is_replace=is_replace_controller, equalizer=eq)
