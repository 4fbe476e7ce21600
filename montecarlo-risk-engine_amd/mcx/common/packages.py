"""Global dtype / host-device contract (mirrors the reference's common/packages.py:1-9: FLOAT = float64).

`device` is the HOST device used for the tiny parameter tensors and closed-form helpers; bulk arrays (paths, exposures)
live on the GPU and are allocated by the native backend (mcx._native)."""
import torch

FLOAT = torch.float64
device = torch.device("cpu")
