"""sparsify_amd: the SAE / transcoder train step of EleutherAI/sparsify on MI355X (gfx950).

Python keeps the reference's operator surface; the compute is libsparsify_hip.so
(include/sparsify_hip.h), bound with ctypes. Importing the package does not need a GPU;
calling any op does, and fails loudly without the built library."""
from .config import SaeConfig, SparseCoderConfig, TrainConfig, TranscoderConfig
from .fused_encoder import EncoderOutput, FusedEncoder, fused_encoder
from .sign_sgd import SignSGD
from .sparse_coder import ForwardOutput, Sae, SparseCoder
from .utils import decoder_impl, hip_decode
from .trainer import Trainer
from .driver import ModelTrainer

__all__ = [
    "EncoderOutput", "ForwardOutput", "FusedEncoder", "ModelTrainer", "Sae", "SaeConfig", "SignSGD", "SparseCoder",
    "SparseCoderConfig", "TrainConfig", "Trainer", "TranscoderConfig", "decoder_impl", "fused_encoder", "hip_decode",
]
