// pybind11 module `rspmm` with the reference extension's export names and Tensor signatures
// (/root/reference/ultra/rspmm/source/rspmm.cpp:256-283, rspmm.h:22-105), as a thin shim over the C ABI of
// libultra_amd.so (include/ultra_rspmm.h).  With this translation unit in place of the reference's
// source/rspmm.cpp + rspmm.cu, the unchanged Python wrapper ultra/rspmm/rspmm.py:182-208 (cpp_extension.load(
// "rspmm", ...)) loads the MI355X engine: its dispatch on `input.device.type == "cuda"` (rspmm.py:20-23) lands on the
// rspmm_<sum>_<mul>_{forward,backward}_cuda names below.  No kernel code here: the shared library is opened at import
// (ULTRA_AMD_LIB, or lib/libultra_amd.so beside / above this module) and every call forwards plain pointers.
//
// The `_cpu` names exist because the reference module has them (rspmm.cpp:256-269); the engine has no CPU path and
// they raise.
#include <dlfcn.h>
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <string>
#include <tuple>

#include "../../../include/ultra_rspmm.h"

namespace {

using at::Tensor;

void *g_lib = nullptr;

void *engine() {
    if (g_lib) return g_lib;
    std::string tried;
    const char *env = std::getenv("ULTRA_AMD_LIB");
    if (env) {
        g_lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        tried += std::string(env) + " ";
    }
    if (!g_lib) {
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(&engine), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            dir = dir.substr(0, dir.find_last_of('/'));
            for (const char *rel : {"/libultra_amd.so", "/lib/libultra_amd.so", "/../lib/libultra_amd.so"}) {
                const std::string p = dir + rel;
                tried += p + " ";
                if ((g_lib = dlopen(p.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
            }
        }
    }
    TORCH_CHECK(g_lib, "rspmm: libultra_amd.so not found (tried: ", tried, "); set ULTRA_AMD_LIB");
    return g_lib;
}

template <typename F>
F symbol(const char *name) {
    void *s = dlsym(engine(), name);
    TORCH_CHECK(s, "rspmm: libultra_amd.so does not export ", name);
    return reinterpret_cast<F>(s);
}

const char *last_error() {
    static auto fn = symbol<const char *(*)()>("ultra_last_error");
    return fn();
}

// rspmm_forward_check / rspmm_backward_check, rspmm.cpp:15-38
void check_forward(const Tensor &edge_index, const Tensor &edge_type, const Tensor &edge_weight, const Tensor &relation,
                   const Tensor &input) {
    TORCH_CHECK(edge_index.dim() == 2 && edge_index.size(0) == 2, "Expect `edge_index` to be (2, num_edge)");
    TORCH_CHECK(edge_type.dim() == 1 && edge_weight.dim() == 1 && relation.dim() == 2 && input.dim() == 2,
                "Expect 1-dimensional edge_type / edge_weight and 2-dimensional relation / input");
    TORCH_CHECK(edge_index.scalar_type() == at::kLong && edge_type.scalar_type() == at::kLong,
                "Expect int64 `edge_index` and `edge_type`");
    TORCH_CHECK(edge_weight.scalar_type() == relation.scalar_type() && relation.scalar_type() == input.scalar_type(),
                "Expect edge_weight, relation and input of the same type");
    TORCH_CHECK(input.scalar_type() == at::kFloat || input.scalar_type() == at::kDouble, "rspmm supports float32 / float64");
    TORCH_CHECK(edge_type.size(0) == edge_index.size(1) && edge_weight.size(0) == edge_index.size(1),
                "Expect edge_type and edge_weight of size (num_edge,)");
    TORCH_CHECK(relation.size(1) == input.size(1), "Expect relation.size(1) == input.size(1)");
    for (const Tensor *t : {&edge_index, &edge_type, &edge_weight, &relation, &input})
        TORCH_CHECK(t->is_cuda() && t->get_device() == input.get_device(), "Expect all tensors on the same GPU");   // checkAllSameGPU
}

typedef int32_t (*forward_fn)(const int64_t *, const int64_t *, const void *, const void *, const void *, void *, int64_t, int64_t,
                              int64_t, int64_t, int32_t, void *);
typedef int32_t (*backward_fn)(const int64_t *, const int64_t *, const void *, const void *, const void *, const void *, const void *,
                               void *, void *, void *, int64_t, int64_t, int64_t, int64_t, int32_t, void *);

Tensor forward(const char *sym, const Tensor &edge_index_, const Tensor &edge_type_, const Tensor &edge_weight_,
               const Tensor &relation_, const Tensor &input_) {
    check_forward(edge_index_, edge_type_, edge_weight_, relation_, input_);
    c10::DeviceGuard guard(input_.device());       // cudaSetDevice(input.get_device()), rspmm.cu:243
    const Tensor edge_index = edge_index_.contiguous(), edge_type = edge_type_.contiguous(), edge_weight = edge_weight_.contiguous(),
                 relation = relation_.contiguous(), input = input_.contiguous();
    Tensor output = at::empty_like(input);
    const int rc = symbol<forward_fn>(sym)(edge_index.data_ptr<int64_t>(), edge_type.data_ptr<int64_t>(), edge_weight.data_ptr(),
                                           relation.data_ptr(), input.data_ptr(), output.data_ptr(), edge_index.size(1),
                                           input.size(0), relation.size(0), input.size(1),
                                           input.scalar_type() == at::kFloat ? ULTRA_F32 : ULTRA_F64,
                                           c10::hip::getCurrentHIPStream(input.get_device()).stream());
    TORCH_CHECK(rc == ULTRA_OK, last_error());
    return output;
}

std::tuple<Tensor, Tensor, Tensor> backward(const char *sym, const Tensor &edge_index_, const Tensor &edge_type_,
                                            const Tensor &edge_weight_, const Tensor &relation_, const Tensor &input_,
                                            const Tensor &output_, const Tensor &output_grad_) {
    check_forward(edge_index_, edge_type_, edge_weight_, relation_, input_);
    TORCH_CHECK(output_.sizes() == input_.sizes() && output_grad_.sizes() == input_.sizes(),
                "Expect output and output_grad of the size of input");
    c10::DeviceGuard guard(input_.device());
    const Tensor edge_index = edge_index_.contiguous(), edge_type = edge_type_.contiguous(), edge_weight = edge_weight_.contiguous(),
                 relation = relation_.contiguous(), input = input_.contiguous(), output = output_.contiguous(),
                 output_grad = output_grad_.contiguous();
    Tensor weight_grad = at::zeros_like(edge_weight), relation_grad = at::zeros_like(relation), input_grad = at::zeros_like(input);
    const int rc = symbol<backward_fn>(sym)(edge_index.data_ptr<int64_t>(), edge_type.data_ptr<int64_t>(), edge_weight.data_ptr(),
                                            relation.data_ptr(), input.data_ptr(), output.data_ptr(), output_grad.data_ptr(),
                                            weight_grad.data_ptr(), relation_grad.data_ptr(), input_grad.data_ptr(),
                                            edge_index.size(1), input.size(0), relation.size(0), input.size(1),
                                            input.scalar_type() == at::kFloat ? ULTRA_F32 : ULTRA_F64,
                                            c10::hip::getCurrentHIPStream(input.get_device()).stream());
    TORCH_CHECK(rc == ULTRA_OK, last_error());
    return std::make_tuple(weight_grad, relation_grad, input_grad);
}

Tensor no_cpu_forward(const Tensor &, const Tensor &, const Tensor &, const Tensor &, const Tensor &) {
    TORCH_CHECK(false, "rspmm: this build is the MI355X engine; it has no CPU path");
}
std::tuple<Tensor, Tensor, Tensor> no_cpu_backward(const Tensor &, const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                   const Tensor &, const Tensor &) {
    TORCH_CHECK(false, "rspmm: this build is the MI355X engine; it has no CPU path");
}

}  // namespace

#define ULTRA_BIND(SUM, MUL)                                                                                                   \
    m.def("rspmm_" #SUM "_" #MUL "_forward_cpu", &no_cpu_forward);                                                             \
    m.def("rspmm_" #SUM "_" #MUL "_backward_cpu", &no_cpu_backward);                                                           \
    m.def("rspmm_" #SUM "_" #MUL "_forward_cuda",                                                                              \
          [](const Tensor &ei, const Tensor &et, const Tensor &ew, const Tensor &rel, const Tensor &in) {                      \
              return forward("ultra_rspmm_" #SUM "_" #MUL "_forward_cuda", ei, et, ew, rel, in);                               \
          });                                                                                                                  \
    m.def("rspmm_" #SUM "_" #MUL "_backward_cuda", [](const Tensor &ei, const Tensor &et, const Tensor &ew, const Tensor &rel, \
                                                       const Tensor &in, const Tensor &out, const Tensor &og) {                \
        return backward("ultra_rspmm_" #SUM "_" #MUL "_backward_cuda", ei, et, ew, rel, in, out, og);                          \
    });

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    ULTRA_BIND(add, mul)
    ULTRA_BIND(min, mul)
    ULTRA_BIND(max, mul)
    ULTRA_BIND(add, add)
    ULTRA_BIND(min, add)
    ULTRA_BIND(max, add)
}
