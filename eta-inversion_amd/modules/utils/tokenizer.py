"""Tokenizer access for the prompt tables.  The reference uses the CLIP BPE tokenizer of the loaded pipeline
(modules/utils/seq_aligner.py:114-115, ptp_utils.py:313).  When a local snapshot is configured (`ETAINV_SD_PATH`) the
byte-level BPE below runs on its `tokenizer/vocab.json` + `merges.txt`; otherwise a word-level stand-in is used (one token per
whitespace word), which is what synthetic-weight runs and the golden fixtures use (the CLIP vocabulary is not in the image)."""
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


def _bytes_to_unicode():
    """the printable stand-ins of the 256 byte values used by GPT-2 / CLIP byte-level BPE (published table)"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class ClipBPETokenizer:
    """CLIP's byte-level BPE tokenizer ([3P] transformers `CLIPTokenizer`, used through the pipeline at reference
    modules/inversion/diffusion_inversion.py:223-241, seq_aligner.py:114-115, ptp_utils.py:313) restated on the published
    algorithm: lower-case + whitespace clean, the CLIP pre-tokenisation pattern, bytes -> printable symbols, greedy lowest-rank
    merges with the `</w>` end-of-word marker, `<|startoftext|>` / `<|endoftext|>` framing, padding with `<|endoftext|>`.
    Reads `vocab.json` + `merges.txt` of a diffusers snapshot; pinned against the installed transformers implementation on a
    synthetic vocabulary (tests/test_host_logic.py)."""
    model_max_length = 77
    _PAT = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""

    def __init__(self, vocab_file, merges_file):
        import json
        import regex
        with open(vocab_file, encoding="utf-8") as f:
            self.encoder = json.load(f)
        self.decoder = {v: k for k, v in self.encoder.items()}
        with open(merges_file, encoding="utf-8") as f:
            lines = f.read().strip().split("\n")
        lines = lines[1:] if lines and lines[0].startswith("#") else lines
        self.ranks = {tuple(l.split()): i for i, l in enumerate(lines) if l}
        self.b2u = _bytes_to_unicode()
        self.u2b = {v: k for k, v in self.b2u.items()}
        self.pat = regex.compile(self._PAT, regex.IGNORECASE)
        self.bos_token_id, self.eos_token_id = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self.pad_token_id = self.eos_token_id
        self._cache = {}

    def _bpe(self, token):
        if token in self._cache:
            return self._cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word, word[1:]))
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        self._cache[token] = word
        return word

    def tokenize(self, text):
        text = " ".join(text.strip().split()).lower()
        out = []
        for tok in self.pat.findall(text):
            out.extend(self._bpe("".join(self.b2u[b] for b in tok.encode("utf-8"))))
        return out

    def encode(self, text):
        unk = self.eos_token_id
        return [self.bos_token_id] + [self.encoder.get(t, unk) for t in self.tokenize(text)] + [self.eos_token_id]

    def decode(self, ids):
        if hasattr(ids, "tolist"):
            ids = ids.tolist()
        if isinstance(ids, int):
            ids = [ids]
        text = "".join(self.decoder[int(i)] for i in ids)
        raw = bytearray(self.u2b[c] for c in text.replace("</w>", " ") if c in self.u2b)
        return raw.decode("utf-8", errors="replace").strip() if not text.startswith("<|") else text.replace("</w>", " ").strip()

    def __call__(self, texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt"):
        import torch
        if isinstance(texts, str):
            texts = [texts]
        rows = []
        for t in texts:
            ids = self.encode(t)
            if truncation and len(ids) > max_length:
                ids = ids[:max_length - 1] + [self.eos_token_id]
            rows.append(ids + [self.pad_token_id] * (max_length - len(ids)))

        class _Out:
            input_ids = torch.tensor(rows, dtype=torch.int64)
        return _Out()


def load_tokenizer():
    """CLIP BPE tokenizer of the snapshot named by ETAINV_SD_PATH.  The word-level stand-in is only for runs WITHOUT a snapshot
    (synthetic weights): real CLIP / UNet weights fed with stand-in token ids would edit silently wrong, so a snapshot without its
    tokenizer files is an error."""
    path = os.environ.get("ETAINV_SD_PATH")
    if not path:
        return WordLevelTokenizer()
    vocab, merges = os.path.join(path, "tokenizer", "vocab.json"), os.path.join(path, "tokenizer", "merges.txt")
    for f in (vocab, merges):
        if not os.path.isfile(f):
            raise FileNotFoundError(f"ETAINV_SD_PATH={path} is set but {f} is missing: the snapshot's text encoder needs its own BPE tokenizer")
    return ClipBPETokenizer(vocab, merges)
