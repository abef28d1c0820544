"""MI355X-native implementation of the reference's `diff_gaussian_rasterization` Python API.

Mirrors DGR/diff_gaussian_rasterization/__init__.py (DGR = gaussian_splatting/submodules/
diff-gaussian-rasterization): same names, same 12-field settings tuple (:157-169), same `forward` keyword
arguments and exceptions (:187-195), same returns `(color[3,H,W] float32, radii[P] int32)`, same backward
gradient order (:143-155), same debug-dump behaviour (:83-90,:132-139).  Underneath, `_C` is a thin ctypes
binding of the C ABI in include/sugar_raster.h (hand-written HIP kernels for gfx950) playing the role of the
reference's pybind module (DGR/ext.cpp:15-19, DGR/rasterize_points.cu:35-217).

There is no CPU path: tensors must live on a ROCm device and the HIP library must be built.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from .. import _lib


# Optional gradient sinks (extension, not part of the reference API): inside `with grad_sink(means3D=buf, shs=buf)` the
# backward of every rasterizer call made in the block writes dL/dmeans3D and dL/dshs straight into the given buffers and
# returns those buffers as the gradients, instead of allocating fresh tensors.  The train step points them at its flat
# gradient buffer, so the 192 MB SH gradient is produced in place and never copied or accumulated.
# `grad_sink(compact_sh=True, out=holder)` selects the compact SH mode of sgr_backward: no SH gradient is produced (the
# autograd gradient of `shs` is None) and holder["masked_colors"] receives the clamp-masked dL/dRGB [P,3] from which
# sugar_amd.train_step rebuilds the SH gradient summed over all views (sgr_sh_grad_from_views).
# `grad_sink(binning_capacity=n, header_out=pinned int32[16], header_event=torch.cuda.Event)` selects the sync-free forward
# (sgr_forward_ex): no host round trip for num_rendered; the caller checks the header (a backward on an invalid forward is a
# no-op on the device).  `tile_need=` / `tile_need_out=` (int32[tiles] device tensors): the walk hint of sgr_forward_opts;
# `tile_order=` / `tile_order_out=` (int32[tiles]): its launch order (deepest tiles first, written by the previous visit).
# `speculative=False`: this call takes the plain host round trip for num_rendered instead of the speculative forward (below).
# `single_level_binning=True`: SGR_FLAG_SINGLE_LEVEL_BINNING.  `dens_stats=(max_radii2D, grad_accum, denom)` (float[P]
# device tensors): the densification statistics of train.py:111-123 fused into the backward (sgr_backward_opts).
_GRAD_SINK: dict = {}

# Speculative forward (SGR_FLAG_SPECULATIVE, the default for callers that ask for nothing else): the reference's forward waits for
# num_rendered in the middle of the pipeline (rasterizer_impl.cu:280-281) with the GPU idle.  From the second call with the same
# (device, P, W, H) on, the forward is enqueued whole with a list capacity of 1.5 x the largest count seen so far, and the one wait --
# for the header the tile scan writes -- happens at the END of the call, with the list pass and the blend kernel queued behind it.
# The call still returns the true num_rendered; a count beyond the capacity costs a repeat of the two tail kernels, never a wrong
# image.  SGR_SPECULATIVE_FORWARD=0 switches it off process-wide.
import os as _os
_SPECULATE = _os.environ.get("SGR_SPECULATIVE_FORWARD", "1") != "0"
_SPEC_CAP: dict = {}  # (device index, P, W, H) -> largest num_rendered seen


def _spec_capacity(key):
    r = _SPEC_CAP.get(key)
    return 0 if not r else r + r // 2 + 65536


# The PyTorch C++ extension (csrc/torch_ext.cpp -> sugar_amd/_C_ext.so, built by sugar_amd.build.build_torch_ext): the reference's
# `_C` module as a torch extension -- same three functions, same argument orders (DGR/ext.cpp:15-19) -- taking tensors and the current
# stream straight to the C ABI.  The plain reference-shaped call (no grad_sink extension in effect) goes through it: no ctypes
# marshalling of ~35 arguments per call, no Python re-entry for the three scratch allocations.  SGR_TORCH_EXT=0 keeps everything on
# the ctypes binding; a missing extension does the same (the HIP library itself is never optional).
_EXT = None
_EXT_TRIED = False
_PLAIN_SINK_KEYS = frozenset(("speculative",))


def _ext():
    global _EXT, _EXT_TRIED
    if _EXT_TRIED:
        return _EXT
    _EXT_TRIED = True
    if _os.environ.get("SGR_TORCH_EXT", "1") == "0" or _os.environ.get("SGR_LIB_PATH"):
        return None   # (SGR_LIB_PATH: an A/B variant of the library is loaded through ctypes; the extension is linked to the default one)
    path = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "_C_ext.so")
    if not _os.path.exists(path):
        return None
    import importlib.util
    _lib.load()  # (the extension links libsugar_raster.so by name: make sure it is THIS tree's copy that is already mapped)
    spec = importlib.util.spec_from_file_location("_C_ext", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.abi_version() != _lib.ABI_VERSION:
        raise ImportError(f"{path}: built against ABI {mod.abi_version()}, bindings are {_lib.ABI_VERSION} (rebuild: python -m sugar_amd.build)")
    _EXT = mod
    return mod


@contextlib.contextmanager
def grad_sink(**buffers):
    global _GRAD_SINK
    old = _GRAD_SINK
    _GRAD_SINK = {k: v for k, v in buffers.items() if v is not None}
    try:
        yield
    finally:
        _GRAD_SINK = old


def _sink_or_empty(sink, name, shape, **f):
    t = sink.get(name) if sink else None
    if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == f["dtype"] and t.device == f["device"] and t.is_contiguous():
        return t
    return torch.empty(*shape, **f)


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def _ptr(t: torch.Tensor | None):
    """Device pointer of a tensor, or NULL for the reference's "absent input" convention: unused optional
    inputs arrive as empty (CPU) tensors (DGR/diff_gaussian_rasterization/__init__.py:197-207)."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _dev_f32(t: torch.Tensor, device, name: str) -> torch.Tensor:
    if t.numel() == 0:
        return t
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device} but means3D is on {device}: the HIP rasterizer needs all "
                           "inputs on the same ROCm device (there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _bucket(nbytes: int) -> int:
    """next of eight sizes per octave (multiples of 2^(floor(log2 n) - 3)), at least 1 MiB granularity above 8 MiB"""
    if nbytes <= (1 << 20):
        return nbytes
    step = 1 << (nbytes.bit_length() - 4)
    return (nbytes + step - 1) // step * step


class _Scratch:
    """The three opaque scratch tensors (geomBuffer, binningBuffer, imgBuffer of rasterize_points.cu:73-78),
    handed to the C ABI through allocation callbacks."""

    def __init__(self, device):
        self.device = device
        self.tensors = {}
        self._cbs = {}

    def cb(self, name: str):
        def alloc(_user, nbytes):
            # The instance list is sized by num_rendered, which differs from camera to camera: exact sizes make PyTorch's caching
            # allocator see a new size nearly every call and go back to hipMalloc for it (19.5 ms per forward at 6M Gaussians @ 4K
            # against 1.7 ms of kernels).  Requests are rounded up to eight sizes per octave (at most 12.5 % more memory) so that a
            # handful of cached blocks serves every view.
            t = torch.empty(_bucket(int(nbytes)), dtype=torch.uint8, device=self.device)
            self.tensors[name] = t
            return t.data_ptr()

        fn = _lib.ALLOC_FN(alloc)
        self._cbs[name] = fn
        return fn

    def release(self):
        """The tensors, with the object's reference cycle (self -> callbacks -> closures -> self) broken: left to the cyclic
        collector, every call's scratch (1.8 GB at 6M Gaussians @ 4K) stayed allocated until a collection happened to run, the
        caching allocator went to hipMalloc for the next call's buffers, and the reference-shaped forward took 19-27 ms instead
        of 1.6 (bench.py --workload config5 --host-sync)."""
        t, self.tensors = self.tensors, {}
        self._cbs.clear()
        return t


class _CModule:
    """ctypes stand-in for the reference's pybind module `_C` (DGR/ext.cpp:15-19)."""

    last_forward: dict = {}

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                            prefiltered, debug):
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
        if not means3D.is_cuda:
            raise RuntimeError("the HIP rasterizer needs tensors on a ROCm device (got CPU tensors); "
                               "there is no CPU fallback")
        lib = _lib.load()
        dev = means3D.device
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        sink0 = _GRAD_SINK or {}
        ext = _ext() if (P != 0 and _PLAIN_SINK_KEYS.issuperset(sink0)) else None
        if ext is not None:
            # the plain reference-shaped call: through the torch C++ extension
            spec_key = (dev.index, P, W, H)
            cap = _spec_capacity(spec_key) if (_SPECULATE and sink0.get("speculative", True)) else 0
            rendered, out_color, radii, geom, binning, img = ext.rasterize_gaussians_ex(
                int(cap), background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix,
                projmatrix, float(tan_fovx), float(tan_fovy), H, W, sh, int(degree), campos, bool(prefiltered), bool(debug))
            mode, sync_free, speculation, list_cap = ext.last_forward_info()
            if _SPECULATE and int(rendered) > _SPEC_CAP.get(spec_key, 0):
                _SPEC_CAP[spec_key] = int(rendered)
            _CModule.last_forward = dict(num_rendered=int(rendered), W=W, H=H, P=P, geom=geom, binning=binning, img=img,
                                         binning_mode=int(mode), sync_free=bool(sync_free), speculative=cap > 0,
                                         speculation_missed=int(speculation) == 2, list_capacity=int(list_cap), torch_ext=True)
            return int(rendered), out_color, radii, geom, binning, img
        if P == 0:  # rasterize_points.cu:68-69,81
            empty = torch.empty(0, dtype=torch.uint8, device=dev)
            return (0, torch.zeros(3, H, W, dtype=torch.float32, device=dev),
                    torch.zeros(0, dtype=torch.int32, device=dev), empty, empty.clone(), empty.clone())
        means3D = _dev_f32(means3D, dev, "means3D")
        background = _dev_f32(background, dev, "bg")
        colors = _dev_f32(colors, dev, "colors_precomp")
        opacity = _dev_f32(opacity, dev, "opacities")
        scales = _dev_f32(scales, dev, "scales")
        rotations = _dev_f32(rotations, dev, "rotations")
        cov3D_precomp = _dev_f32(cov3D_precomp, dev, "cov3D_precomp")
        viewmatrix = _dev_f32(viewmatrix, dev, "viewmatrix")
        projmatrix = _dev_f32(projmatrix, dev, "projmatrix")
        sh = _dev_f32(sh, dev, "shs")
        campos = _dev_f32(campos, dev, "campos")
        M = sh.size(1) if sh.numel() != 0 else 0  # rasterize_points.cu:83-87
        out_color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        scratch = _Scratch(dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            # sync-free forward: `binning_capacity=n` (instances); the 8-word device header is copied into the pinned
            # tensor `header_out` behind the forward and `header_event` recorded: the CALLER must check it (word 0 = real
            # num_rendered <= n and word 6 == 0) before running the backward, and repeat the forward otherwise
            sink = _GRAD_SINK or {}
            capacity = int(sink.get("binning_capacity") or 0)
            spec_key = (dev.index, P, W, H)
            speculate = (_SPECULATE and capacity == 0 and sink.get("speculative", True) and not sink.get("single_level_binning")
                         and sink.get("tile_need") is None)
            if speculate:
                capacity = _spec_capacity(spec_key)
                speculate = capacity > 0
            # `raw_params=True`: scales / rotations / opacities are the raw 3DGS parameters, activated inside the kernels
            flags = (_lib.SGR_FLAG_RAW_PARAMS if sink.get("raw_params") else 0) | \
                    (_lib.SGR_FLAG_SINGLE_LEVEL_BINNING if sink.get("single_level_binning") else 0) | \
                    (_lib.SGR_FLAG_SPECULATIVE if speculate else 0)
            hdr_out, hdr_ev = sink.get("header_out"), sink.get("header_event")
            if capacity > 0 and not speculate and (hdr_out is None or hdr_ev is None):
                raise RuntimeError("binning_capacity needs header_out (pinned int32[16]) and header_event (torch.cuda.Event)")
            if hdr_out is not None and not (hdr_out.is_pinned() and hdr_out.numel() >= 16 and hdr_out.dtype == torch.int32):
                raise RuntimeError("header_out must be a pinned int32 tensor of 16 elements")
            need, need_out = sink.get("tile_need"), sink.get("tile_need_out")
            order, order_out = sink.get("tile_order"), sink.get("tile_order_out")
            info = _lib.ForwardInfo()
            opts = _lib.ForwardOpts(capacity, flags, hdr_out.data_ptr() if hdr_out is not None else None, None,
                                    need.data_ptr() if need is not None else None,
                                    need_out.data_ptr() if need_out is not None else None, float(sink.get("hint_margin") or 0.0),
                                    int(sink.get("chunk_grid") or 0), C.pointer(info),
                                    order.data_ptr() if order is not None else None,
                                    order_out.data_ptr() if order_out is not None else None)
            rendered = lib.sgr_forward_ex(
                scratch.cb("geom"), None, scratch.cb("binning"), None, scratch.cb("img"), None,
                P, int(degree), int(M), _ptr(background), W, H, _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity),
                _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
                _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                _ptr(out_color), _ptr(radii), int(bool(debug)), C.c_void_p(stream), C.byref(opts))
            if rendered >= 0 and hdr_ev is not None:
                # the library has queued the header copies (words 0-7 behind the tile scan, 8-15 behind the blend)
                hdr_ev.record(torch.cuda.current_stream(dev))
        if rendered < 0:
            raise RuntimeError(f"sgr_forward failed ({rendered}): {_lib.last_error()}")
        if _SPECULATE and (speculate or not info.sync_free):  # (the count is the true one in both cases)
            if int(rendered) > _SPEC_CAP.get(spec_key, 0):
                _SPEC_CAP[spec_key] = int(rendered)
        t = scratch.release()
        # introspection only (bench.py's roofline accounting, parity tests): the most recent forward's scratch
        _CModule.last_forward = dict(num_rendered=int(rendered), W=W, H=H, P=P, geom=t["geom"], binning=t["binning"],
                                     img=t["img"], binning_mode=int(info.binning_mode), sync_free=bool(info.sync_free),
                                     speculative=bool(speculate), speculation_missed=int(info.speculation) == 2,
                                     list_capacity=capacity if (speculate and int(info.speculation) == 1) else int(rendered), torch_ext=False)
        return int(rendered), out_color, radii, t["geom"], t["binning"], t["img"]

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh,
                                     degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, grad_out=None):
        lib = _lib.load()
        dev = means3D.device
        P = means3D.size(0)
        if not grad_out and P != 0:
            ext = _ext()
            if ext is not None:  # the plain call: through the torch C++ extension
                return ext.rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, float(scale_modifier),
                                                        cov3D_precomp, viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy),
                                                        dL_dout_color, sh, int(degree), campos, geomBuffer, int(R), binningBuffer,
                                                        imageBuffer, bool(debug))
        H, W = dL_dout_color.size(1), dL_dout_color.size(2)
        M = sh.size(1) if sh.dim() == 3 else 0  # (the reference takes 0 for an empty tensor and then fails in autograd when P == 0)
        f = dict(dtype=torch.float32, device=dev)
        use_cov = cov3D_precomp.numel() != 0
        _p = lambda t: None if t is None else _ptr(t)
        # Every row of these is written by sgr_backward, so no 300 MB of zero-fill per call
        # (the reference allocates nine torch::zeros, rasterize_points.cu:151-159).
        dL_dmeans3D = _sink_or_empty(grad_out, "means3D", (P, 3), **f)
        # `grad_sink(params_only=True)`: dL/dmeans2D, dL/dconic and dL/dcov3D are not produced (None gradients)
        lean = bool(grad_out and grad_out.get("params_only")) and not use_cov
        dL_dmeans2D = None if lean else torch.empty(P, 3, **f)
        dL_dcolors = _sink_or_empty(grad_out, "colors", (P, 3), **f)
        dL_dconic = None if lean else torch.empty(P, 2, 2, **f)
        dL_dopacity = _sink_or_empty(grad_out, "opacities", (P, 1), **f)
        dL_dcov3D = None if lean else torch.empty(P, 6, **f)
        compact_sh = bool(grad_out and grad_out.get("compact_sh")) and M > 0
        dL_dsh = None if compact_sh else _sink_or_empty(grad_out, "shs", (P, M, 3), **f)
        if dL_dsh is not None and colors.numel() != 0 and dL_dsh.numel() != 0:
            dL_dsh.zero_()  # SHs AND precomputed colours: the kernels take the colours and never touch dL_dsh (reference: zeros)
        dL_dscales = torch.zeros(P, 3, **f) if use_cov else _sink_or_empty(grad_out, "scales", (P, 3), **f)
        dL_drotations = torch.zeros(P, 4, **f) if use_cov else _sink_or_empty(grad_out, "rotations", (P, 4), **f)
        raw_mode = 4 if (grad_out and grad_out.get("raw_params")) else 0  # SGR_MODE_RAW_PARAMS
        if compact_sh and grad_out.get("sh_dir_elsewhere") and sh is not None and sh.numel() > 0:
            raw_mode |= 8  # SGR_MODE_SH_DIR_ELSEWHERE: dL_dmeans3D lacks the view-direction term (sgr_sh_adam_from_views_ex forms it)
        if P != 0:
            means3D = _dev_f32(means3D, dev, "means3D")
            dL = _dev_f32(dL_dout_color, dev, "dL_dout_color")
            colors = _dev_f32(colors, dev, "colors_precomp"); scales = _dev_f32(scales, dev, "scales")
            rotations = _dev_f32(rotations, dev, "rotations"); sh = _dev_f32(sh, dev, "shs")
            cov3D_precomp = _dev_f32(cov3D_precomp, dev, "cov3D_precomp")
            # (contiguous copies must stay referenced until the call has been enqueued: SuGaR hands over a TRANSPOSED view matrix,
            # sugar_model.py:2143-2144 -- a temporary's block would go back to the allocator before its pointer is used)
            background = _dev_f32(background, dev, "bg")
            viewmatrix = _dev_f32(viewmatrix, dev, "viewmatrix")
            projmatrix = _dev_f32(projmatrix, dev, "projmatrix")
            campos = _dev_f32(campos, dev, "campos")
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                args = (P, int(degree), int(M), int(R), _ptr(background), W, H, _ptr(means3D), _ptr(sh), _ptr(colors),
                        _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
                        _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), _ptr(radii),
                        _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(dL),
                        _p(dL_dmeans2D), _p(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dmeans3D),
                        _p(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations), int(bool(debug)),
                        C.c_void_p(stream))
                on_colors = grad_out.get("on_colors") if (compact_sh and grad_out) else None
                stats = grad_out.get("dens_stats") if grad_out else None
                bopts = None
                # (the forward of this call sorted the tiles when it was given tile_order_out: no second sort)
                bflags = _lib.SGR_BWD_TILE_ORDER_READY if (grad_out and grad_out.get("tile_order_out") is not None) else 0
                if stats is not None:
                    for st in stats:
                        if not (st.is_cuda and st.dtype == torch.float32 and st.numel() == P and st.is_contiguous()):
                            raise RuntimeError("dens_stats: three contiguous float32 device tensors of P elements")
                    bopts = C.byref(_lib.BackwardOpts(*[st.data_ptr() for st in stats], None, bflags))
                elif bflags:
                    bopts = C.byref(_lib.BackwardOpts(None, None, None, None, bflags))
                if on_colors is not None:
                    # two halves: the masked colour gradients are final after the blend half, so the caller can start
                    # exchanging them while the preprocess half runs
                    rc = lib.sgr_backward_ex(1 | raw_mode, *args, bopts)
                    if rc >= 0:
                        on_colors(dL_dcolors)
                        rc = lib.sgr_backward_ex(2 | raw_mode, *args, bopts)
                else:
                    rc = lib.sgr_backward_ex(raw_mode, *args, bopts)
            if rc < 0:
                raise RuntimeError(f"sgr_backward failed ({rc}): {_lib.last_error()}")
        if compact_sh and isinstance(grad_out.get("out"), dict):
            grad_out["out"]["masked_colors"] = dL_dcolors
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        lib = _lib.load()
        if not means3D.is_cuda:
            raise RuntimeError("the HIP rasterizer needs tensors on a ROCm device; there is no CPU fallback")
        dev = means3D.device
        P = means3D.size(0)
        present = torch.zeros(P, dtype=torch.bool, device=dev)
        if P != 0:
            means3D = _dev_f32(means3D, dev, "means3D")
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                rc = lib.sgr_mark_visible(P, _ptr(means3D), _ptr(_dev_f32(viewmatrix, dev, "viewmatrix")),
                                          _ptr(_dev_f32(projmatrix, dev, "projmatrix")), _ptr(present), C.c_void_p(stream))
            if rc < 0:
                raise RuntimeError(f"sgr_mark_visible failed ({rc}): {_lib.last_error()}")
        return present


_C = _CModule


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        # argument order of the C++ entry point, DGR/rasterize_points.h:18-38
        args = (raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy,
                raster_settings.image_height, raster_settings.image_width, sh, raster_settings.sh_degree,
                raster_settings.campos, raster_settings.prefiltered, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.grad_sink = dict(_GRAD_SINK)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        args = (raster_settings.bg, means3D, radii, colors_precomp, scales, rotations, raster_settings.scale_modifier,
                cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.tanfovx,
                raster_settings.tanfovy, grad_out_color, sh, raster_settings.sh_degree, raster_settings.campos,
                geomBuffer, num_rendered, binningBuffer, imgBuffer, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
                 grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(*args, grad_out=ctx.grad_sink)
        # gradient order of DGR/diff_gaussian_rasterization/__init__.py:143-155
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings)
