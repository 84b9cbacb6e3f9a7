from .multibit_embedder import AWAREEmbedder

__all__ = ["AWAREEmbedder"]
