"""Per-clip watermark embedding (batched adversarial optimisation on the GPU)."""
from . import losses, optimizers, schedulers
from .multibit_embedder import AWAREEmbedder

__all__ = ["AWAREEmbedder", "losses", "optimizers", "schedulers"]
