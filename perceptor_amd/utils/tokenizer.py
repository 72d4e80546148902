"""Byte-pair tokenizer of OpenAI CLIP (host-side string work in front of the text tower, engine/text.py).

``open_clip.tokenize`` (called at perceptor/models/open_clip.py:101-103) and ``transformers.CLIPTokenizer``
(perceptor/models/stable_diffusion/stable_diffusion.py:298-311) implement the same published scheme: lower-cased, whitespace-collapsed
text is split by a fixed pattern, each piece is mapped byte-wise to printable code points, and adjacent symbols are merged greedily in
the rank order of a merge list; ids are [start] + pieces + [end], zero-padded (open_clip) or end-padded (transformers) to the context
length.  The merge list (``bpe_simple_vocab_16e6.txt.gz``, 48 894 merges) is DATA that ships with CLIP checkpoints and is not part of
this repository: pass its path, or set ``PERCEPTOR_AMD_BPE``.  ``ftfy`` (mojibake repair in front of the tokenizer upstream) is not
available here and is skipped: it is the identity on well-formed text.
"""
from __future__ import annotations

import gzip
import html
import os
from typing import Dict, List, Optional, Sequence, Tuple

import regex
import torch

_N_MERGES = 49152 - 256 - 2
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
_SPECIAL = ("<|startoftext|>", "<|endoftext|>")


def _byte_alphabet() -> List[str]:
    """Code point standing for each byte value 0..255: printable Latin-1 bytes map to themselves, the rest to 256, 257, ..."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spare = [], 256
    for b in range(256):
        if b in keep:
            table.append(chr(b))
        else:
            table.append(chr(spare))
            spare += 1
    return table


def _vocabulary_order(table: Sequence[str]) -> List[str]:
    """The id order of the single-byte symbols: the 188 printable bytes first (ascending), then the remapped ones."""
    printable = [b for b in range(256) if ord(table[b]) == b]
    others = [b for b in range(256) if ord(table[b]) != b]
    return [table[b] for b in printable + others]


class ClipTokenizer:
    def __init__(self, bpe_path: Optional[str] = None, merges: Optional[Sequence[Tuple[str, str]]] = None):
        if merges is None:
            bpe_path = bpe_path or os.environ.get("PERCEPTOR_AMD_BPE")
            if not bpe_path or not os.path.exists(bpe_path):
                raise FileNotFoundError("CLIP merge list not found: pass bpe_path= (bpe_simple_vocab_16e6.txt.gz, shipped with CLIP "
                                        "checkpoints) or set PERCEPTOR_AMD_BPE")
            opener = gzip.open if bpe_path.endswith(".gz") else open
            with opener(bpe_path, "rb") as f:
                lines = f.read().decode("utf-8").split("\n")
            merges = [tuple(l.split()) for l in lines[1:_N_MERGES + 1]]
        self.table = _byte_alphabet()
        symbols = _vocabulary_order(self.table)
        vocab = symbols + [s + "</w>" for s in symbols] + [a + b for a, b in merges] + list(_SPECIAL)
        self.ids: Dict[str, int] = {s: i for i, s in enumerate(vocab)}
        self.rank: Dict[Tuple[str, str], int] = {tuple(m): i for i, m in enumerate(merges)}
        self.sot, self.eot = self.ids[_SPECIAL[0]], self.ids[_SPECIAL[1]]
        self._memo: Dict[str, List[int]] = {}

    @property
    def vocab_size(self) -> int:
        return len(self.ids)

    def _merge(self, piece: str) -> List[int]:
        """Greedy lowest-rank-first merging of one pre-token (already in the byte alphabet)."""
        if piece in self._memo:
            return self._memo[piece]
        if piece in _SPECIAL:
            out = [self.ids[piece]]
        else:
            sym = list(piece[:-1]) + [piece[-1] + "</w>"]
            while len(sym) > 1:
                best, best_rank = None, None
                for a, b in zip(sym, sym[1:]):
                    r = self.rank.get((a, b))
                    if r is not None and (best_rank is None or r < best_rank):
                        best, best_rank = (a, b), r
                if best is None:
                    break
                merged, i = [], 0
                while i < len(sym):        # every non-overlapping occurrence of the pair, left to right
                    if i + 1 < len(sym) and sym[i] == best[0] and sym[i + 1] == best[1]:
                        merged.append(best[0] + best[1])
                        i += 2
                    else:
                        merged.append(sym[i])
                        i += 1
                sym = merged
            out = [self.ids[s] for s in sym]
        self._memo[piece] = out
        return out

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        out: List[int] = []
        for piece in _SPLIT.findall(text):
            out.extend(self._merge("".join(self.table[b] for b in piece.encode("utf-8"))))
        return out

    def __call__(self, texts, context_length: int = 77, pad: str = "zero") -> torch.Tensor:
        """texts -> int64 [N, context_length].  pad="zero": open_clip.tokenize (zeros behind the end token; over-long prompts are cut and
        their last id set to the end token); pad="eot": transformers' CLIPTokenizer(padding="max_length", truncation=True)."""
        if isinstance(texts, str):
            texts = [texts]
        fill = 0 if pad == "zero" else self.eot
        out = torch.full((len(texts), context_length), fill, dtype=torch.int64)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out
