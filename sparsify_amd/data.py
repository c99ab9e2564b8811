"""Token data for the on-the-fly activation driver: GPT-style
chunking of a text dataset into rows of exactly `max_seq_len` token ids (sparsify/data.py:14-122) and
the reference's memory-mapped token file format (sparsify/data.py:125-157 -- a flat array of token
ids, `ctx_len` per row). CPU preprocessing in front of the path: the tokenizer is the caller's
(transformers), the dataset a `datasets.Dataset`; nothing here touches the GPU."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset


def _rows_of_batch(texts: list[str], tokenizer, row_len: int, keep_tail: bool) -> list[list[int]]:
    """One batch of documents -> rows of `row_len` token ids. The documents are joined into one string,
    an end-of-text token in front of each (so every row stream starts with one), tokenized in one call
    and cut into rows; the tail that does not fill a row is dropped unless `keep_tail`."""
    sep = tokenizer.eos_token or "<|endoftext|>"
    enc = tokenizer(sep + sep.join(texts), max_length=row_len, truncation=True, return_overflowing_tokens=True,
                    return_attention_mask=False)
    ids = enc["input_ids"]
    if ids and isinstance(ids[0], int):
        # a slow (pure-Python) tokenizer: the first row, and everything behind it as one flat list
        rest = enc.get("overflowing_tokens") or []
        rows = [ids] + [rest[i: i + row_len] for i in range(0, len(rest), row_len)]
    else:
        rows = list(ids)                  # a fast tokenizer has cut the rows itself
    if not keep_tail:
        rows = rows[:-1]                  # (the last row is almost never full, and rows are never padded)
    if not rows:
        raise ValueError("Not enough data to create a single complete batch."
                         " Either allow the final batch to be returned, or supply more data.")
    return rows


def chunk_and_tokenize(data, tokenizer, *, format: str = "torch", num_proc: int | None = None, text_key: str = "text",
                       max_seq_len: int = 2048, return_final_batch: bool = False, load_from_cache_file: bool = True):
    """sparsify/data.py:14-101 for a `datasets.Dataset` / `DatasetDict`: every row of the result is
    `min(tokenizer.model_max_length, max_seq_len)` token ids (`input_ids`, all other columns dropped);
    long documents span several rows, short ones share a row with their neighbours, separated by the
    end-of-text token. Documents are processed 2048 at a time and each such batch gives up its unfilled
    last row, like the reference (`num_proc` defaults to half the cores, as there)."""
    import os

    row_len = min(tokenizer.model_max_length, max_seq_len)
    columns = data.column_names
    if isinstance(columns, dict):         # DatasetDict: the splits must agree (data.py:104-122)
        per_split = list(columns.values())
        if any(c != per_split[0] for c in per_split):
            raise ValueError("All splits must have the same columns")
        columns = per_split[0]
    if num_proc is None:
        num_proc = max(1, (os.cpu_count() or 2) // 2)
    out = data.map(lambda batch: {"input_ids": _rows_of_batch(batch[text_key], tokenizer, row_len, return_final_batch)},
                   batched=True, batch_size=2048, num_proc=num_proc if num_proc > 1 else None, remove_columns=columns,
                   load_from_cache_file=load_from_cache_file)
    return out.with_format(format, columns=["input_ids"])


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
