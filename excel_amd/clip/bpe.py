"""CLIP's byte-level BPE tokenizer (the published OpenAI CLIP algorithm; counterpart of clip/simple_tokenizer.py and of
`tokenize`, clip/clip.py:206-248).  Host-side, one-time (text-bank building); needs CLIP's merges file
`bpe_simple_vocab_16e6.txt.gz` (not shipped here: pass its path, or set EXCEL_BPE_VOCAB).

Vocabulary layout (49 408 entries): 256 byte symbols, the same 256 with the end-of-word mark, 48 894 merges in file
order, <|startoftext|>, <|endoftext|>.
"""
import gzip
import html
import os

import numpy as np

try:                       # `regex` understands \p{L}; the stdlib `re` does not
    import regex as _re
except ImportError:        # pragma: no cover
    _re = None

EOW = "</w>"
SOT, EOT = "<|startoftext|>", "<|endoftext|>"
N_MERGES = 49152 - 256 - 2


def _byte_symbols():
    """Printable stand-ins for the 256 byte values (GPT-2 convention): printable latin-1 bytes map to themselves, the rest
    to code points from 256 upwards, in byte order."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, nxt = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + nxt)
            nxt += 1
    ordered = keep + [b for b in range(256) if b not in keep]      # vocabulary order: kept bytes first, then the remapped ones
    return table, [table[b] for b in ordered]


class BPETokenizer:
    def __init__(self, vocab_path=None):
        vocab_path = vocab_path or os.environ.get("EXCEL_BPE_VOCAB")
        if not vocab_path or not os.path.exists(vocab_path):
            raise FileNotFoundError("BPETokenizer needs CLIP's bpe_simple_vocab_16e6.txt.gz (pass vocab_path= or set EXCEL_BPE_VOCAB)")
        if _re is None:
            raise RuntimeError("BPETokenizer needs the `regex` module (\\p{L} classes)")
        lines = gzip.open(vocab_path).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        self.byte_sym, base = _byte_symbols()
        vocab = base + [s + EOW for s in base] + ["".join(m) for m in merges] + [SOT, EOT]
        self.encoder = {s: i for i, s in enumerate(vocab)}
        self.decoder = {i: s for s, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.cache = {SOT: SOT, EOT: EOT}
        self.splitter = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                    _re.IGNORECASE)

    # one word (already mapped to byte symbols) -> space-joined BPE pieces
    def _merge(self, word):
        if word in self.cache:
            return self.cache[word]
        parts = list(word[:-1]) + [word[-1] + EOW]
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            out, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and (parts[i], parts[i + 1]) == best:
                    out.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    out.append(parts[i])
                    i += 1
            parts = out
        res = " ".join(parts)
        self.cache[word] = res
        return res

    @staticmethod
    def _clean(text):
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    def encode(self, text):
        ids = []
        for tok in self.splitter.findall(self._clean(text)):
            sym = "".join(self.byte_sym[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge(sym).split(" "))
        return ids

    def decode(self, ids):
        inv = {s: b for b, s in self.byte_sym.items()}
        words = "".join(self.decoder[int(i)] for i in ids).split(EOW)
        return " ".join(bytearray(inv[c] for c in w if c in inv).decode("utf-8", errors="replace") for w in words)


def tokenize(texts, tokenizer, context_length=77, truncate=False):
    """clip/clip.py:206-248 -> int32 [n, context_length] (numpy)."""
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = tokenizer.encoder[SOT], tokenizer.encoder[EOT]
    out = np.zeros((len(texts), context_length), np.int32)
    for i, t in enumerate(texts):
        ids = [sot] + tokenizer.encode(t) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, :len(ids)] = ids
    return out
