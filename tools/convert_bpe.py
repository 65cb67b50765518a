"""Convert the OpenAI CLIP byte-BPE merge list (MIT-licensed data shipped with CLIP as
bpe_simple_vocab_16e6.txt.gz) into the id-pair table spn4cir_amd/tokenizer.py loads.

    python tools/convert_bpe.py <path/to/bpe_simple_vocab_16e6.txt.gz> spn4cir_amd/assets/clip_bpe_merges.npz

Output: int32 [48894, 2]: merge r joins vocabulary ids (a, b) into id 512 + r.  Ids 0..255 are the 256
byte symbols (in the byte->printable order CLIP uses), 256..511 the same symbols carrying the end-of-word
marker; 49406 / 49407 are <|startoftext|> / <|endoftext|>."""
import gzip
import sys

import numpy as np


def byte_symbols():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return bs, [chr(c) for c in cs]


def main(src, dst):
    lines = gzip.open(src).read().decode("utf-8").split("\n")
    merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1]]
    _, chars = byte_symbols()
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": 256 + i for i, c in enumerate(chars)})
    table = np.zeros((len(merges), 2), dtype=np.int32)
    for r, (a, b) in enumerate(merges):
        table[r] = (vocab[a], vocab[b])
        vocab[a + b] = 512 + r
    assert len(vocab) == 49406, len(vocab)
    np.savez_compressed(dst, merges=table)
    print(dst, table.shape)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
