"""Token data for the on-the-fly activation driver: the reference's memory-mapped token file
format (sparsify/data.py:125-157 -- a flat array of token ids, `ctx_len` per row). Tokenising
raw text (`chunk_and_tokenize`) stays with the caller's tokenizer; it is outside the path."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset


class MemmapDataset(Dataset):
    """Rows of `ctx_len` token ids from a flat binary file, read through a memory map."""

    def __init__(self, data_path: str, ctx_len: int, max_examples: int | None = None, dtype=np.uint16):
        tokens = np.memmap(data_path, dtype=dtype, mode="r")
        rows = tokens[: (tokens.shape[0] // ctx_len) * ctx_len].reshape(-1, ctx_len)
        self.mmap = rows[:max_examples]

    @classmethod
    def _view(cls, rows) -> "MemmapDataset":
        out = cls.__new__(cls)
        out.mmap = rows
        return out

    def __len__(self):
        return len(self.mmap)

    def __getitem__(self, idx):
        return dict(input_ids=torch.from_numpy(np.asarray(self.mmap[idx]).astype(np.int64)))

    def select(self, rng: range) -> "MemmapDataset":
        """Rows rng.start .. rng.stop (resume skips the consumed prefix, trainer.py:272-276)."""
        return self._view(self.mmap[rng.start: rng.stop])

    def shard(self, num_shards: int, shard_id: int) -> "MemmapDataset":
        """The shard_id-th of num_shards nearly equal contiguous parts (one per rank)."""
        return self._view(np.array_split(self.mmap, num_shards)[shard_id])
