"""Writes tests/golden/embeddings_synth.json.gz: a synthetic stand-in for the reference's
data/datasets/R2R_VLNCE_v1-3_preprocessed/embeddings.json.gz (instruction_encoder.py:52-61: a gzipped
JSON list of rows; PAD = row 0 = zeros, UNK = row 1 = the mean of the word rows).  Values are rounded
to 4 decimals so that the JSON text -> float32 path is the same everywhere.

    python tests/golden/make_goldens_embeddings.py
"""
import gzip
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS, DIM = 320, 50
g = torch.Generator().manual_seed(7)
table = (torch.randn(ROWS, DIM, generator=g) * 0.4).mul(1e4).round().div(1e4)
table[0] = 0.0
table[1] = table[2:].mean(0).mul(1e4).round().div(1e4)
with gzip.GzipFile(os.path.join(HERE, "embeddings_synth.json.gz"), "wb", mtime=0) as f:
    f.write(json.dumps([[round(float(v), 4) for v in row] for row in table.tolist()]).encode())
print("wrote", ROWS, "x", DIM)
