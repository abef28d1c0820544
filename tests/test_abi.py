"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares, and the
Python mirror of the reference API keeps its contract (names, argument validation, error behaviour).  No compute calls:
there is no GPU here."""
import ctypes
import glob
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(sgr_[A-Za-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if n != "sgr_alloc_fn"))


def _declared_struct(name):
    """field names of `typedef struct name { ... } name;` in include/sugar_raster.h, in order"""
    src = open(os.path.join(ROOT, "include", "sugar_raster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    fields = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        for part in stmt.split(","):
            fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    return fields


def test_ctypes_structs_mirror_the_header():
    from sugar_amd import _lib
    for cname, cls in (("sgr_forward_opts", _lib.ForwardOpts), ("sgr_forward_info", _lib.ForwardInfo),
                       ("sgr_backward_opts", _lib.BackwardOpts), ("sgr_train_config", _lib.TrainConfig),
                       ("sgr_train_view", _lib.TrainView), ("sgr_train_exchange", _lib.TrainExchange)):
        assert [f[0] for f in cls._fields_] == _declared_struct(cname), cname


def test_python_constants_match_the_header():
    """every SGR_FLAG_* / SGR_BWD_* constant the Python mirror defines has the header's value"""
    from sugar_amd import _lib
    text = open(os.path.join(ROOT, "include", "sugar_raster.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(SGR_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", text)}
    mirrored = [n for n in dir(_lib) if n.startswith(("SGR_FLAG_", "SGR_BWD_"))]
    assert len(mirrored) >= 3
    for n in mirrored:
        assert n in defs and defs[n] == getattr(_lib, n), n
    assert defs["SGR_ABI_VERSION"] == _lib.ABI_VERSION


def test_library_exports_every_declared_symbol(hip_lib):
    raw = ctypes.CDLL(os.path.join(ROOT, "sugar_amd", "libsugar_raster.so"))
    decl = _declared_functions()
    assert len(decl) >= 18
    missing = [n for n in decl if not hasattr(raw, n)]
    assert not missing, missing
    from sugar_amd import _lib
    assert sorted(_lib.SIGNATURES) == decl, "ctypes signature table and header disagree"
    assert hip_lib.sgr_abi_version() == _lib.ABI_VERSION == 4


def test_scratch_layout_queries(hip_lib):
    L = hip_lib
    assert L.sgr_geom_bytes(1000) >= 48 * 1000 and L.sgr_geom_bytes(1000) % 256 == 0
    W, H = 1920, 1080
    T = 120 * 68
    offs = [L.sgr_img_final_T_offset(W, H), L.sgr_img_n_contrib_offset(W, H), L.sgr_img_tile_start_offset(W, H),
            L.sgr_img_tile_maxc_offset(W, H), L.sgr_img_tile_walked_offset(W, H)]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert offs[1] - offs[0] >= 4 * W * H and L.sgr_img_bytes(W, H) >= offs[-1] + 4 * T
    assert L.sgr_binning_point_list_offset(1000) % 256 == 0 and L.sgr_binning_bytes(1000, 64, 64) >= 4000
    assert L.sgr_binning_bytes(0, 64, 64) > 0
    assert L.sgr_geom_bytes(1000) >= 96 * 1000 + 16 * 1000  # records + backward accumulators + depth-sort scratch


def test_loader_fails_loudly_without_library(monkeypatch, tmp_path):
    from sugar_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_settings_tuple_matches_reference_field_order():
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def _rasterizer():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    e = torch.eye(4)
    return GaussianRasterizer(GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, e, e, 3, torch.zeros(3),
                                                            False, False))


def test_argument_validation_matches_reference():
    """DGR/diff_gaussian_rasterization/__init__.py:191-195"""
    r = _rasterizer()
    m = torch.zeros(4, 3); o = torch.zeros(4, 1)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, shs=torch.zeros(4, 16, 3), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, colors_precomp=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, colors_precomp=m, scales=m, rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m, o, colors_precomp=m, scales=m)  # rotations missing


def test_cpu_tensors_are_rejected_not_silently_computed():
    r = _rasterizer()
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(m, m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.markVisible(m)
    with pytest.raises(RuntimeError, match=r"\(num_points, 3\)"):
        r(torch.zeros(4, 2), m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under sugar_amd/ or the drop-in package may reference it."""
    bad = []
    for base in ("sugar_amd", "diff_gaussian_rasterization", "simple_knn"):
        for path in glob.glob(os.path.join(ROOT, base, "**", "*.py"), recursive=True):
            src = open(path).read()
            if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                bad.append(path)
    assert not bad, bad


def test_torch_extension_has_the_reference_module_surface(hip_lib):
    """sugar_amd/_C_ext.so (csrc/torch_ext.cpp): the reference's pybind module `_C` (DGR/ext.cpp:15-19) as a PyTorch C++ extension
    over the C ABI -- the three entry points by the reference's names, built against this ABI version; no CPU path."""
    import pytest
    import torch
    from sugar_amd import build
    build.build_torch_ext()
    import sugar_amd.diff_gaussian_rasterization as dgr
    ext = dgr._ext()
    assert ext is not None and ext.abi_version() == hip_lib.sgr_abi_version()
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):
        assert callable(getattr(ext, name))
    e = torch.empty(0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.rasterize_gaussians(torch.zeros(3), torch.zeros(5, 3), e, torch.zeros(5, 1), torch.ones(5, 3), torch.zeros(5, 4), 1.0, e,
                                torch.eye(4), torch.eye(4), 0.5, 0.5, 16, 16, torch.zeros(5, 16, 3), 3, torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="num_points, 3"):
        ext.rasterize_gaussians(torch.zeros(3), torch.zeros(5, 2), e, torch.zeros(5, 1), torch.ones(5, 3), torch.zeros(5, 4), 1.0, e,
                                torch.eye(4), torch.eye(4), 0.5, 0.5, 16, 16, torch.zeros(5, 16, 3), 3, torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.mark_visible(torch.zeros(5, 3), torch.eye(4), torch.eye(4))
