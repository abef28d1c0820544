// torch_ext.cpp -- the PyTorch-ROCm C++ extension `_C` of the drop-in rasterizer module.
//
// Same three entry points, argument orders and return tuples as the reference's pybind module (DGR/ext.cpp:15-19,
// DGR/rasterize_points.h:18-67, implemented in DGR/rasterize_points.cu:35-217):
//     rasterize_gaussians(...)          -> (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer)
//     rasterize_gaussians_backward(...) -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
//     mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]
// so that the reference's own `diff_gaussian_rasterization/__init__.py` could import it unchanged.  It is plumbing: tensors in,
// raw pointers and the current HIP stream out to the C ABI of include/sugar_raster.h (the hand-written gfx950 kernels of
// libsugar_raster.so, linked by name), scratch tensors handed out through the three allocation callbacks like the reference's
// `resizeFunctional` lambdas (rasterize_points.cu:27-33).  What it does NOT do as the reference does: the gradient tensors are not
// zero-filled (sgr_backward writes every row: 300 MB of memset per call at 1M Gaussians, rasterize_points.cu:151-159), and the
// forward waits for num_rendered at the END of the call instead of in the middle (SGR_FLAG_SPECULATIVE, see the header) from the
// second call with the same (device, P, W, H) on.  No CPU path: CPU tensors raise.
#include <torch/extension.h>
// PyTorch-ROCm presents its devices as DeviceType::CUDA: the guard and the current stream come from the "masquerading" variants
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../include/sugar_raster.h"

namespace {

struct Scratch { torch::Tensor t; c10::Device dev; };
char* alloc_cb(void* user, size_t bytes)
{
    Scratch* s = static_cast<Scratch*>(user);
    // eight sizes per octave above 1 MiB: a handful of cached blocks serves every view (see sugar_amd/diff_gaussian_rasterization)
    size_t n = bytes;
    if (n > (size_t(1) << 20)) {
        int top = 63 - __builtin_clzll((unsigned long long)n);
        const size_t step = size_t(1) << (top - 3);
        n = (n + step - 1) / step * step;
    }
    s->t = torch::empty({(long long)n}, torch::TensorOptions().dtype(torch::kByte).device(s->dev));
    return reinterpret_cast<char*>(s->t.data_ptr());
}

const float* fptr(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

torch::Tensor dev_f32(const torch::Tensor& t, const c10::Device& dev, const char* name)
{
    if (t.numel() == 0) return t;
    TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), " but means3D is on ", dev,
                ": the HIP rasterizer needs all inputs on the same ROCm device (there is no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
    return t.contiguous();
}

// speculation state: largest num_rendered seen per (device, P, W, H); the forward's report for the Python side's introspection
std::mutex g_mu;
std::map<std::tuple<int, int, int, int>, long long> g_seen;
struct LastInfo { int binning_mode = 0, sync_free = 0, speculation = 0; long long capacity = 0; } g_last;
bool g_speculate = true;

}  // namespace

// `capacity_in`: < 0 -- this module's own speculation (what a caller of the reference-shaped entry point gets); 0 -- the plain host
// round trip; > 0 -- speculate with exactly this list capacity (sugar_amd.diff_gaussian_rasterization keeps the guess itself)
static std::tuple<int64_t, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeImpl(long long capacity_in, const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                   const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
                   const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                   const float tan_fovx, const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                   const int degree, const torch::Tensor& campos, const bool prefiltered, const bool debug)
{
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");  // rasterize_points.cu:57-59
    TORCH_CHECK(means3D.is_cuda(), "the HIP rasterizer needs tensors on a ROCm device (got CPU tensors); there is no CPU fallback");
    const c10::Device dev = means3D.device();
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
    auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev);
    auto u8 = torch::TensorOptions().dtype(torch::kByte).device(dev);
    if (P == 0)  // rasterize_points.cu:68-69,81
        return std::make_tuple((int64_t)0, torch::zeros({3, H, W}, f32), torch::zeros({0}, i32), torch::empty({0}, u8), torch::empty({0}, u8),
                               torch::empty({0}, u8));
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const auto m3 = dev_f32(means3D, dev, "means3D"), bg = dev_f32(background, dev, "bg"), col = dev_f32(colors, dev, "colors_precomp"),
               op = dev_f32(opacity, dev, "opacities"), sc = dev_f32(scales, dev, "scales"), ro = dev_f32(rotations, dev, "rotations"),
               cov = dev_f32(cov3D_precomp, dev, "cov3D_precomp"), vm = dev_f32(viewmatrix, dev, "viewmatrix"),
               pm = dev_f32(projmatrix, dev, "projmatrix"), shs = dev_f32(sh, dev, "shs"), cam = dev_f32(campos, dev, "campos");
    const int M = shs.numel() ? (int)shs.size(1) : 0;  // rasterize_points.cu:83-87
    torch::Tensor out_color = torch::empty({3, H, W}, f32);
    torch::Tensor radii = torch::empty({P}, i32);
    Scratch geom{torch::Tensor(), dev}, binning{torch::Tensor(), dev}, img{torch::Tensor(), dev};
    const auto key = std::make_tuple((int)dev.index(), P, W, H);
    long long capacity = capacity_in > 0 ? capacity_in : 0;
    if (capacity_in < 0 && g_speculate) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_seen.find(key);
        if (it != g_seen.end() && it->second > 0) capacity = it->second + it->second / 2 + 65536;
    }
    sgr_forward_info info = {0, 0, 0};
    sgr_forward_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    opts.binning_capacity = capacity;
    opts.flags = capacity > 0 ? SGR_FLAG_SPECULATIVE : 0;
    opts.info = &info;
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    const int64_t rendered = sgr_forward_ex(alloc_cb, &geom, alloc_cb, &binning, alloc_cb, &img, P, degree, M, fptr(bg), W, H, fptr(m3),
                                            fptr(shs), fptr(col), fptr(op), fptr(sc), scale_modifier, fptr(ro), fptr(cov), fptr(vm),
                                            fptr(pm), fptr(cam), tan_fovx, tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(),
                                            radii.data_ptr<int>(), debug ? 1 : 0, (void*)stream, &opts);
    if (rendered < 0) throw std::runtime_error(std::string("sgr_forward failed (") + std::to_string(rendered) + "): " + sgr_last_error());
    {
        std::lock_guard<std::mutex> lk(g_mu);
        long long& seen = g_seen[key];
        if (rendered > seen) seen = rendered;
        g_last.binning_mode = info.binning_mode; g_last.sync_free = info.sync_free; g_last.speculation = info.speculation;
        g_last.capacity = info.speculation == 1 ? capacity : rendered;
    }
    auto def = [&](torch::Tensor& t) { return t.defined() ? t : torch::empty({0}, u8); };
    return std::make_tuple((int64_t)rendered, out_color, radii, def(geom.t), def(binning.t), def(img.t));
}

// (num_rendered and R travel as 64-bit integers: a view can hold up to 2^32 - 2 instances; Python does not see the difference)
std::tuple<int64_t, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                   const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
                   const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                   const float tan_fovx, const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                   const int degree, const torch::Tensor& campos, const bool prefiltered, const bool debug)
{
    return RasterizeImpl(-1, background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                         tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                           const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                           const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                           const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const torch::Tensor& dL_dout_color,
                           const torch::Tensor& sh, const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer,
                           const int64_t R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug)
{
    const c10::Device dev = means3D.device();
    const int P = (int)means3D.size(0), H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const int M = sh.dim() == 3 ? (int)sh.size(1) : 0;
    auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
    const bool use_cov = cov3D_precomp.numel() != 0;
    TORCH_CHECK(means3D.is_cuda(), "the HIP rasterizer needs tensors on a ROCm device (got CPU tensors); there is no CPU fallback");
    // the scratch the forward returned and its radii: same device, the dtypes the forward produced, dense (the C ABI takes raw pointers)
    if (P != 0) {
        TORCH_CHECK(radii.device() == dev && radii.scalar_type() == torch::kInt32 && radii.is_contiguous() && radii.numel() == P,
                    "radii must be the forward's int32[P] tensor on ", dev);
        const torch::Tensor* bufs[3] = {&geomBuffer, &binningBuffer, &imageBuffer};
        const char* names[3] = {"geomBuffer", "binningBuffer", "imageBuffer"};
        for (int k = 0; k < 3; k++)
            TORCH_CHECK(bufs[k]->device() == dev && bufs[k]->scalar_type() == torch::kByte && bufs[k]->is_contiguous() && bufs[k]->numel() > 0,
                        names[k], " must be the forward's uint8 scratch tensor on ", dev, " (contiguous)");
        TORCH_CHECK((size_t)geomBuffer.numel() >= sgr_geom_bytes(P), "geomBuffer is smaller than the forward's geometry scratch for ", P, " Gaussians");
        TORCH_CHECK((size_t)imageBuffer.numel() >= sgr_img_bytes(W, H), "imageBuffer is smaller than the forward's image scratch for ", W, "x", H);
        TORCH_CHECK(dL_dout_color.dim() == 3 && dL_dout_color.size(0) == 3, "dL_dout_color must be [3, H, W]");
    }
    // every row of these is written by sgr_backward (rows of culled Gaussians are set to zero): no zero-fill -- except dL_dsh when
    // the colours were precomputed: the kernels then never touch it, and the reference hands back zeros (rasterize_points.cu:151-159)
    const bool sh_written = M != 0 && colors.numel() == 0;
    torch::Tensor dL_dmeans3D = torch::empty({P, 3}, f32), dL_dmeans2D = torch::empty({P, 3}, f32), dL_dcolors = torch::empty({P, 3}, f32),
                  dL_dconic = torch::empty({P, 2, 2}, f32), dL_dopacity = torch::empty({P, 1}, f32), dL_dcov3D = torch::empty({P, 6}, f32),
                  dL_dsh = sh_written ? torch::empty({P, M, 3}, f32) : torch::zeros({P, M, 3}, f32);
    torch::Tensor dL_dscales = use_cov ? torch::zeros({P, 3}, f32) : torch::empty({P, 3}, f32);
    torch::Tensor dL_drotations = use_cov ? torch::zeros({P, 4}, f32) : torch::empty({P, 4}, f32);
    if (P != 0) {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const auto m3 = dev_f32(means3D, dev, "means3D"), bg = dev_f32(background, dev, "bg"), col = dev_f32(colors, dev, "colors_precomp"),
                   sc = dev_f32(scales, dev, "scales"), ro = dev_f32(rotations, dev, "rotations"),
                   cov = dev_f32(cov3D_precomp, dev, "cov3D_precomp"), vm = dev_f32(viewmatrix, dev, "viewmatrix"),
                   pm = dev_f32(projmatrix, dev, "projmatrix"), shs = dev_f32(sh, dev, "shs"), cam = dev_f32(campos, dev, "campos"),
                   dL = dev_f32(dL_dout_color, dev, "dL_dout_color");
        hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        const int rc = sgr_backward(P, degree, M, (int64_t)R, fptr(bg), W, H, fptr(m3), fptr(shs), fptr(col), fptr(sc), scale_modifier, fptr(ro),
                                    fptr(cov), fptr(vm), fptr(pm), fptr(cam), tan_fovx, tan_fovy, radii.data_ptr<int>(),
                                    reinterpret_cast<char*>(geomBuffer.data_ptr()), reinterpret_cast<char*>(binningBuffer.data_ptr()),
                                    reinterpret_cast<char*>(imageBuffer.data_ptr()), fptr(dL), dL_dmeans2D.data_ptr<float>(),
                                    dL_dconic.data_ptr<float>(), dL_dopacity.data_ptr<float>(), dL_dcolors.data_ptr<float>(),
                                    dL_dmeans3D.data_ptr<float>(), dL_dcov3D.data_ptr<float>(), M ? dL_dsh.data_ptr<float>() : nullptr,
                                    dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(), debug ? 1 : 0, (void*)stream);
        if (rc < 0) throw std::runtime_error(std::string("sgr_backward failed (") + std::to_string(rc) + "): " + sgr_last_error());
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

torch::Tensor markVisible(const torch::Tensor& means3D, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix)
{
    TORCH_CHECK(means3D.is_cuda(), "the HIP rasterizer needs tensors on a ROCm device; there is no CPU fallback");
    const c10::Device dev = means3D.device();
    const int P = (int)means3D.size(0);
    torch::Tensor present = torch::zeros({P}, torch::TensorOptions().dtype(torch::kBool).device(dev));
    if (P != 0) {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const auto m3 = dev_f32(means3D, dev, "means3D"), vm = dev_f32(viewmatrix, dev, "viewmatrix"), pm = dev_f32(projmatrix, dev, "projmatrix");
        hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        const int rc = sgr_mark_visible(P, fptr(m3), fptr(vm), fptr(pm), reinterpret_cast<uint8_t*>(present.data_ptr<bool>()), (void*)stream);
        if (rc < 0) throw std::runtime_error(std::string("sgr_mark_visible failed: ") + sgr_last_error());
    }
    return present;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("rasterize_gaussians", &RasterizeGaussians);
    m.def("rasterize_gaussians_ex", &RasterizeImpl);   // (first argument: the list capacity to speculate with, see RasterizeImpl)
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackward);
    m.def("mark_visible", &markVisible);
    // introspection for the Python side (not part of the reference's module)
    m.def("last_forward_info", []() {
        std::lock_guard<std::mutex> lk(g_mu);
        return std::make_tuple(g_last.binning_mode, g_last.sync_free, g_last.speculation, g_last.capacity);
    });
    m.def("set_speculative", [](bool on) { g_speculate = on; });
    m.def("forget", [](int device, int P, int W, int H) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_seen.erase(std::make_tuple(device, P, W, H));
    });
    m.def("set_seen", [](int device, int P, int W, int H, long long n) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_seen[std::make_tuple(device, P, W, H)] = n;
    });
    m.def("abi_version", []() { return sgr_abi_version(); });
}
