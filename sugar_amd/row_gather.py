"""`x[idx]` with a HIP backward, for the per-Gaussian tensors of a SuGaR model.

The regulariser of the coarse trainers gathers per-Gaussian rows by the million -- `self.points[random_indices]`,
`self.quaternions[...]`, `self.scaling[...]` (sugar_model.py:922-925), `sugar.get_normals()[closest_gaussians_idx]` with 1M x 16
indices (coarse_sdf.py:690-692) -- and autograd's backward of each is a scatter-add that stock PyTorch runs as sort + segmented
reduction: 46 % of the GPU time of an SDF iteration on an MI355X once the rasterizer, the density field, `ssim` and Adam are HIP.
`sgr_scatter_add_rows` (csrc/field.hip) does it with a stable radix grouping of the entries by row and sixteen lanes per row.

Nothing in the reference is edited: `RowGatherTensor` is a `torch.Tensor` subclass whose only behaviour is that indexing it with an
int64 CUDA tensor goes through `row_gather` (same values; the gradient is a deterministic fixed-order sum -- sixteen strided partial sums per row over the stably grouped entries -- reproducible bit for bit, though not the sequential entry-order sum); every other
operation on it is the plain operation and returns plain tensors.  `sugar_amd.sugar_patch.install_row_gathers` makes the SuGaR
properties `points`, `scaling`, `quaternions` and the method `get_normals` return their usual tensor viewed as this subclass."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

MIN_ROWS = 32768   # below this the stock backward is as fast and its additions are ordered


class _RowGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx):
        ctx.save_for_backward(idx)
        ctx.src_shape = tuple(src.shape)
        return src[idx]

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        shape = ctx.src_shape
        P = shape[0]
        W = 1
        for d in shape[1:]:
            W *= d
        lib = _lib.load()
        dev = grad.device
        g = grad.reshape(-1, W)
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        ix = idx.reshape(-1)
        if not ix.is_contiguous():
            ix = ix.contiguous()
        N = ix.numel()
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.sgr_scatter_add_rows_scratch_bytes(N, P), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.sgr_scatter_add_rows(N, C.c_void_p(ix.data_ptr()), C.c_void_p(g.data_ptr()), W, P, C.c_void_p(out.data_ptr()),
                                          C.c_void_p(scratch.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_scatter_add_rows failed ({rc})")
        return out, None


def supported(src, idx) -> bool:
    if not (isinstance(idx, torch.Tensor) and idx.dtype == torch.int64 and idx.is_cuda and src.is_cuda
            and src.dtype == torch.float32 and 1 <= src.dim() <= 3 and idx.numel() >= MIN_ROWS and idx.numel() < 2 ** 32
            and src.shape[0] < 2 ** 31 and src.requires_grad and torch.is_grad_enabled()):
        return False
    w = 1
    for d in src.shape[1:]:
        w *= d
    return 1 <= w <= 4


def row_gather(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src[idx] for a [P, ...] float32 CUDA tensor (at most four floats per row) and an int64 CUDA index tensor of any shape"""
    return _RowGather.apply(src, idx)


class RowGatherTensor(torch.Tensor):
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        with torch._C.DisableTorchFunctionSubclass():
            if func is torch.Tensor.__getitem__ and len(args) == 2 and not kwargs and isinstance(args[0], RowGatherTensor):
                src = args[0].as_subclass(torch.Tensor)
                if supported(src, args[1]):
                    return _RowGather.apply(src, args[1])
                return src[args[1]]
            return func(*args, **kwargs)


def as_row_gather(t):
    """the same tensor (storage, autograd history) viewed as a RowGatherTensor; anything that is not a float32 CUDA tensor with at
    most four floats per row is returned as it is"""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and 1 <= t.dim() <= 3) or isinstance(t, RowGatherTensor):
        return t
    w = 1
    for d in t.shape[1:]:
        w *= d
    if w > 4:
        return t
    return t.as_subclass(RowGatherTensor)
