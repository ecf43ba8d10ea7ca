"""Portable, counter-based synthetic data and weights (numpy Philox), shared by the parity
tests, the golden-vector generator and bench.py so that every side sees identical bags.

Bags mimic ResNet patch features (non-negative), omic vectors are N(0,1); seed 42 is the
reference's default seed (config/config_mine.yaml:37)."""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(seed: int, tag: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(tag.encode())]))


def normal(shape: Iterable[int], seed: int, tag: str, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    a = _rng(seed, tag).standard_normal(tuple(shape), dtype=np.float32) * np.float32(std) + np.float32(mean)
    return torch.from_numpy(a.astype(np.float32))


def bag(batch: int, n: int, feat: int, seed: int = 42, tag: str = "bag") -> torch.Tensor:
    """[batch, n, feat] non-negative instance features: max(N(0,1) * 0.5, 0)."""
    return torch.clamp(normal((batch, n, feat), seed, tag, std=0.5), min=0.0)


def fill_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 42, tag: str = "w") -> Dict[str, torch.Tensor]:
    """Deterministic weights for a {name: shape} table.  Matrices/filters get N(0, 1/sqrt(fan_in)),
    LayerNorm-style 'norm*.weight' vectors get 1 + 0.1 N(0,1), other vectors 0.1 N(0,1), so that no
    path (bias, affine, offsets) is trivially zero in the parity tests."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = normal(shape, seed, tag + ":" + name, std=1.0 / np.sqrt(max(fan_in, 1)))
        elif "norm" in name and name.endswith("weight"):
            t = normal(shape, seed, tag + ":" + name, std=0.1, mean=1.0)
        else:
            t = normal(shape, seed, tag + ":" + name, std=0.1)
        out[name] = t
    return out
