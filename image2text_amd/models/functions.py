"""Gradient normaliser (reference models/functions.py:4-27).

In the HIP path the rule  g <- g / (||g||_2 + 1e-6)  (norm over the whole (B, T, d) tensor) is applied inside the
hand-written backward at every block output (engine.HotPath._blocks_bwd -> i2t_grad_normalize).  This module keeps
the reference's public name for code that wants the autograd form on an arbitrary GPU tensor.
"""
import torch

from .. import ops


class _NormalizeGradients(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_):
        return input_.view_as(input_)

    @staticmethod
    def backward(ctx, grad_output):
        g = grad_output.to(torch.float32).contiguous().clone()
        ws = torch.zeros(1, dtype=torch.float32, device=g.device)
        ops.grad_normalize(g, ws)
        return g.to(grad_output.dtype)


normalize_gradients = _NormalizeGradients.apply
