"""Small vector helpers the callers of the render path use (mirror of generators/math_utils_torch.py:8-26)."""
import torch


def transform_vectors(matrix: torch.Tensor, vectors4: torch.Tensor) -> torch.Tensor:
    """(M,M) applied to row vectors (N,M) -> (N,M)."""
    return vectors4 @ matrix.T


def normalize_vecs(vectors: torch.Tensor) -> torch.Tensor:
    """v / |v| along the last dim, no epsilon (math_utils_torch.py:16-20)."""
    return vectors / vectors.norm(dim=-1, keepdim=True)


def torch_dot(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return (x * y).sum(-1)
