"""Tokenizer access for the prompt tables.  The reference uses the CLIP BPE tokenizer of the loaded pipeline
(modules/utils/seq_aligner.py:114-115, ptp_utils.py:313).  When a local snapshot is configured (`ETAINV_SD_PATH`) the real
`transformers.CLIPTokenizer` is loaded from it; otherwise a word-level stand-in is used (one token per whitespace word),
which is what synthetic-weight runs and the golden fixtures use."""
import os
import zlib


class WordLevelTokenizer:
    model_max_length = 77
    bos_token_id, eos_token_id = 49406, 49407

    def __init__(self):
        self._words = {self.bos_token_id: "<|startoftext|>", self.eos_token_id: "<|endoftext|>"}

    def encode(self, text):
        ids = [self.bos_token_id]
        for w in text.split(" "):
            if w:
                i = 1000 + zlib.crc32(w.encode()) % 40000
                self._words[i] = w
                ids.append(i)
        return ids + [self.eos_token_id]

    def decode(self, ids):
        return " ".join(self._words[int(i)] for i in ids)

    def __call__(self, texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt"):
        import torch
        rows = []
        for t in texts:
            ids = self.encode(t)[:max_length]
            rows.append(ids + [self.eos_token_id] * (max_length - len(ids)))

        class _Out:
            input_ids = torch.tensor(rows, dtype=torch.int64)
        return _Out()


def load_tokenizer():
    path = os.environ.get("ETAINV_SD_PATH")
    if path and os.path.isdir(os.path.join(path, "tokenizer")):
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer"))
    return WordLevelTokenizer()
