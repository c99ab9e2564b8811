import os
import sys
from pathlib import Path

# The CPU tests are many tiny torch ops; on a box whose vCPUs are shared, 8 OpenMP threads spin on
# each other (measured here: 24 s against 1.2 s on one thread for 200 small steps). Two threads
# unless the environment says otherwise (the GPU box sets OMP_NUM_THREADS itself); set before torch
# loads so that the world-size-2 gloo subprocesses inherit it.
os.environ.setdefault("OMP_NUM_THREADS", "2")

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str) -> dict:
    """Load one committed golden fixture (made by tools/make_golden.py from the reference)."""
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        out = {}
        for k in z.files:
            a = z[k]
            out[k] = torch.from_numpy(a) if a.ndim > 0 else a.item()
        return out


def golden_names(prefix: str) -> list[str]:
    return sorted(p.stem for p in GOLDEN.glob(f"{prefix}*.npz"))


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


TOY_WORDS = ("the of and to in is that it was for on are as with his they at be this from have or by one had not but what "
             "all were when we there can an your which their said if do will each about how up out them then she many some "
             "so these would other into has more her two like him see time could no make than first been its who now "
             "people my made over did down only way find use may water long little very after words called just where").split()


def toy_tokenizer(model_max_length: int = 64):
    """A word-level fast tokenizer over a fixed vocabulary (no download, no training): what
    tools/make_golden.py fed the reference's chunk_and_tokenize and what the test feeds ours."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<|endoftext|>": 0, "<unk>": 1, **{w: i + 2 for i, w in enumerate(TOY_WORDS)}}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    return PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|endoftext|>", unk_token="<unk>",
                                   model_max_length=model_max_length)


def toy_documents(n: int = 300, seed: int = 0) -> list[str]:
    import random
    rng = random.Random(seed)
    docs = []
    for _ in range(n):
        words = [rng.choice(TOY_WORDS) if rng.random() > 0.03 else "zyx" for _ in range(rng.choice([1, 3, 7, 20, 45, 120]))]
        docs.append(" ".join(words))
    return docs
