// `_gridencoder`: the reference's pybind11 operator module as a torch extension over the C ABI of libucnerf_march.so.
//
// ref: /root/reference/nerf/gridencoder/src/bindings.cpp:5-9 (the three exported names), gridencoder.h:12-15 (their
// signatures), gridencoder.cu:15-18, 449-465, 474-496 (the TORCH_CHECK preconditions -> RuntimeError).  Same positional
// arguments; the caller allocates every output (grid.py:47-52, 77-82); kernels go to the CURRENT HIP stream; nothing
// synchronises except the first read of a level-offsets tensor (the C ABI takes the offsets as host metadata; the copy is
// cached per tensor + version counter).  With `ucnerf_amd/compat/native` on sys.path the reference's grid.py:10
// (`import _gridencoder as _backend`) binds to this module unchanged; `ucnerf_amd/compat/_gridencoder.py` is the ctypes
// form of the same three functions for builds without this extension.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <cstdint>
#include <mutex>
#include <vector>

#include "ucnerf_march.h"

namespace {

// Preconditions with the messages of the reference's host functions (gridencoder.cu:15-18, 449-465, 474-496), raised as
// c10::Error -> Python RuntimeError before anything is launched.
enum class Kind { Floating, Int };

void require(const at::Tensor &t, const char *name, Kind kind) {
    TORCH_CHECK(t.device().is_cuda(), name, " must be a CUDA tensor");
    TORCH_CHECK(t.is_contiguous(), name, " must be a contiguous tensor");
    const auto st = t.scalar_type();
    if (kind == Kind::Int) {
        TORCH_CHECK(st == at::ScalarType::Int, name, " must be an int tensor");
    } else {
        TORCH_CHECK(st == at::ScalarType::Float || st == at::ScalarType::Half || st == at::ScalarType::Double, name,
                    " must be a floating tensor");
    }
}
void require(const at::optional<at::Tensor> &t, const char *name) {
    if (t.has_value() && t->defined()) require(*t, name, Kind::Floating);
}

// host copies of level-offset tensors: keyed on the TensorImpl, valid while that impl is alive and unmodified
struct OffsetEntry {
    c10::weak_intrusive_ptr<c10::TensorImpl> impl;
    uint32_t version;
    std::vector<int32_t> host;
};
std::mutex g_mutex;
std::vector<OffsetEntry> g_offsets;

const int32_t *host_offsets(const at::Tensor &offsets, uint32_t L) {
    TORCH_CHECK(offsets.numel() >= (int64_t)L + 1, "offsets must hold L + 1 entries");
    c10::TensorImpl *impl = offsets.unsafeGetTensorImpl();
    const uint32_t version = offsets._version();
    std::lock_guard<std::mutex> lock(g_mutex);
    for (size_t i = 0; i < g_offsets.size();) {
        if (g_offsets[i].impl.expired()) {
            g_offsets.erase(g_offsets.begin() + i);
            continue;
        }
        if (g_offsets[i].impl._unsafe_get_target() == impl && g_offsets[i].version == version) return g_offsets[i].host.data();
        i++;
    }
    const at::Tensor h = offsets.to(at::kCPU).contiguous();
    OffsetEntry e{c10::weak_intrusive_ptr<c10::TensorImpl>(offsets.getIntrusivePtr()), version,
                  std::vector<int32_t>(h.data_ptr<int32_t>(), h.data_ptr<int32_t>() + h.numel())};
    for (size_t i = 0; i < g_offsets.size(); i++)
        if (g_offsets[i].impl._unsafe_get_target() == impl) {
            g_offsets[i] = std::move(e);
            return g_offsets[i].host.data();
        }
    g_offsets.push_back(std::move(e));
    return g_offsets.back().host.data();
}

int dtype_code(const at::Tensor &t, const char *what) {
    if (t.scalar_type() == at::ScalarType::Float) return 0;   // emb_dtype: 0 = float32, 1 = float16 (include/ucnerf_march.h)
    if (t.scalar_type() == at::ScalarType::Half) return 1;
    TORCH_CHECK(false, what, " must be float32 or float16 on this build (float64 tables are not supported)");
    return 0;
}

void check_rc(int rc) { TORCH_CHECK(rc == 0, ucn_last_error()); }

ucn_stream_t current_stream() { return (ucn_stream_t)c10::hip::getCurrentHIPStream().stream(); }

void *opt_ptr(const at::optional<at::Tensor> &t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }

void grid_encode_forward(const at::Tensor inputs, const at::Tensor embeddings, const at::Tensor offsets, at::Tensor outputs,
                         const uint32_t B, const uint32_t D, const uint32_t C, const uint32_t L, const float S, const uint32_t H,
                         at::optional<at::Tensor> dy_dx, const uint32_t gridtype, const bool align_corners, const uint32_t interp) {
    require(inputs, "inputs", Kind::Floating);
    require(embeddings, "embeddings", Kind::Floating);
    require(offsets, "offsets", Kind::Int);
    require(outputs, "outputs", Kind::Floating);
    require(dy_dx, "dy_dx");
    TORCH_CHECK(inputs.scalar_type() == at::ScalarType::Float, "inputs must be float32 (gridencoder.cu:469 reads them as float)");
    TORCH_CHECK(outputs.scalar_type() == embeddings.scalar_type(), "outputs must have the embeddings' dtype");
    check_rc(ucn_grid_encode_forward(inputs.data_ptr<float>(), embeddings.data_ptr(), host_offsets(offsets, L), outputs.data_ptr(), B, D,
                                     C, L, S, H, opt_ptr(dy_dx), gridtype, align_corners ? 1 : 0, interp,
                                     dtype_code(embeddings, "embeddings"), current_stream()));
}

void grid_encode_backward(const at::Tensor grad, const at::Tensor inputs, const at::Tensor embeddings, const at::Tensor offsets,
                          at::Tensor grad_embeddings, const uint32_t B, const uint32_t D, const uint32_t C, const uint32_t L,
                          const float S, const uint32_t H, const at::optional<at::Tensor> dy_dx, at::optional<at::Tensor> grad_inputs,
                          const uint32_t gridtype, const bool align_corners, const uint32_t interp) {
    require(grad, "grad", Kind::Floating);
    require(inputs, "inputs", Kind::Floating);
    require(embeddings, "embeddings", Kind::Floating);
    require(offsets, "offsets", Kind::Int);
    require(grad_embeddings, "grad_embeddings", Kind::Floating);
    require(dy_dx, "dy_dx");
    require(grad_inputs, "grad_inputs");
    TORCH_CHECK(grad.scalar_type() == grad_embeddings.scalar_type(), "grad and grad_embeddings must have the same dtype");
    check_rc(ucn_grid_encode_backward(grad.data_ptr(), inputs.data_ptr<float>(), embeddings.data_ptr(), host_offsets(offsets, L),
                                      grad_embeddings.data_ptr(), B, D, C, L, S, H, opt_ptr(dy_dx), opt_ptr(grad_inputs), gridtype,
                                      align_corners ? 1 : 0, interp, dtype_code(grad, "grad"), current_stream()));
}

void grad_total_variation(const at::Tensor inputs, const at::Tensor embeddings, at::Tensor grad, const at::Tensor offsets,
                          const float weight, const uint32_t B, const uint32_t D, const uint32_t C, const uint32_t L, const float S,
                          const uint32_t H, const uint32_t gridtype, const bool align_corners) {
    require(inputs, "inputs", Kind::Floating);
    require(embeddings, "embeddings", Kind::Floating);
    require(grad, "grad", Kind::Floating);
    require(offsets, "offsets", Kind::Int);
    TORCH_CHECK(inputs.scalar_type() == at::ScalarType::Float && embeddings.scalar_type() == at::ScalarType::Float &&
                    grad.scalar_type() == at::ScalarType::Float,
                "grad_total_variation: float32 tensors only on this build");
    check_rc(ucn_grad_total_variation(inputs.data_ptr<float>(), embeddings.data_ptr<float>(), grad.data_ptr<float>(),
                                      host_offsets(offsets, L), weight, B, D, C, L, S, H, gridtype, align_corners ? 1 : 0,
                                      current_stream()));
}

}  // namespace

PYBIND11_MODULE(_gridencoder, m) {
    m.def("grid_encode_forward", &grid_encode_forward, "hash-grid interpolation, forward (+ dy_dx): HIP kernels of grid_op.hip on the current stream");
    m.def("grid_encode_backward", &grid_encode_backward, "hash-grid interpolation, table (+ input) gradients: HIP kernels of grid_op.hip on the current stream");
    m.def("grad_total_variation", &grad_total_variation, "total-variation gradient of the table, accumulated into grad");
    m.def("abi_version", []() { return ucn_abi_version(); }, "ABI version of the libucnerf_march.so this module is linked to");
}
