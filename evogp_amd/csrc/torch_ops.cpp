// torch_ops.cpp — libtorch registration of the operator boundary: TORCH_LIBRARY(evogp_cuda) with the reference's five
// schemas verbatim (src/evogp/cuda/torch_wrapper.cu:291-299) and their CUDA-key implementations (:301-307; ROCm builds of
// PyTorch dispatch HIP tensors on the CUDA key), plus the extra ops of this engine in the evogp_hip namespace.
//
// Each implementation does what the reference's wrapper does — validate the sizes (torch_wrapper.cu:48-54,103-104,154-156,
// 205-208,250-254), require contiguous device tensors of the exact shape (check_tensor, :7-17), allocate the outputs on the
// device of the designated input (:63,116,168,217,264) — and then calls the C ABI of include/evogp_hip.h with raw device
// pointers on torch's CURRENT stream of that device (the reference launches on the legacy default stream).  Differences:
// dtypes are checked explicitly (the reference relies on data_ptr<T>()), every tensor must live on the same device, and
// launch errors are reported (the reference's check_cuda_error calls are commented out, :84,135,188,231,282).
//
// Host code only (no kernels): compiled with g++ against the torch headers and linked to libevogp_hip.so.  There is no CPU
// implementation and no fallback: a CPU tensor fails in the dispatcher exactly as with the reference.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <hip/hip_runtime_api.h>

#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "../../include/evogp_hip.h"

namespace {

using at::Tensor;
using Tensor3 = std::tuple<Tensor, Tensor, Tensor>;

constexpr int64_t kMaxStack = EVOGP_MAX_STACK;

void check_tensor(const Tensor &t, at::IntArrayRef shape, const char *name, const c10::Device &dev,
                  c10::optional<at::ScalarType> dtype = c10::nullopt) {
    TORCH_CHECK(t.is_cuda() && t.is_contiguous(), name, " must be a contiguous CUDA tensor");
    TORCH_CHECK(t.device() == dev, name, " lives on ", t.device(), " but the call runs on ", dev, ": all operands of one call must share a device");
    TORCH_CHECK(t.sizes() == shape, name, " must have shape ", shape, ", but got shape ", t.sizes());
    if (dtype) TORCH_CHECK(t.scalar_type() == *dtype, "expected scalar type ", *dtype, " for ", name, " but found ", t.scalar_type());
}

void check_forest(int64_t pop, int64_t gp_len, const Tensor &value, const Tensor &type, const Tensor &size, const c10::Device &dev,
                  const char *suffix = "") {
    const std::string v = std::string("value") + suffix, t = std::string("type") + suffix, s = std::string("subtree_size") + suffix;
    check_tensor(value, {pop, gp_len}, v.c_str(), dev, at::kFloat);
    check_tensor(type, {pop, gp_len}, t.c_str(), dev, at::kShort);
    check_tensor(size, {pop, gp_len}, s.c_str(), dev, at::kShort);
}

void check_sizes(int64_t pop_size, int64_t gp_len) {
    TORCH_CHECK(pop_size > 0, "pop_size must be larger than 0, but got ", pop_size);
    TORCH_CHECK(gp_len > 0 && gp_len <= kMaxStack, "gp_len must be in range (0, ", kMaxStack, "], but got ", gp_len);
}

void check_rc(int rc, const char *what) {
    TORCH_CHECK(rc == 0, what, " failed: ", evogp_hip_error_string(rc), " (code ", rc, ")");
}

// torch's current stream of the device.  (ROCm builds of torch present their devices as DeviceType::CUDA; the generic
// c10::DeviceGuard resolves to the registered implementation, and the stream pool is indexed by the device ordinal.)
evogp_stream_t current_stream(const c10::Device &dev) { return (evogp_stream_t)c10::hip::getCurrentHIPStream(dev.index()).stream(); }

Tensor3 empty_forest(int64_t rows, int64_t gp_len, const c10::Device &dev) {
    return {at::empty({rows, gp_len}, at::TensorOptions().dtype(at::kFloat).device(dev)),
            at::empty({rows, gp_len}, at::TensorOptions().dtype(at::kShort).device(dev)),
            at::empty({rows, gp_len}, at::TensorOptions().dtype(at::kShort).device(dev))};
}

// the reference reads keys.data_ptr<unsigned int>() (torch_wrapper.cu:76); signed integer keys carry the same 32-bit patterns
Tensor keys_u32(const Tensor &keys) {
    if (keys.scalar_type() == at::kUInt32) return keys;
    TORCH_CHECK(keys.scalar_type() == at::kInt || keys.scalar_type() == at::kLong, "keys must be uint32/int32/int64, got ", keys.scalar_type());
    return keys.to(at::kLong).bitwise_and(0xFFFFFFFFLL).to(at::kUInt32).contiguous();
}

Tensor3 generate_impl(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, int64_t const_samples_len, double out_prob,
                      double const_prob, const Tensor &keys_in, const Tensor &depth2leaf_probs, const Tensor &roulette_funcs,
                      const Tensor &const_samples, int64_t tree_index_offset, const Tensor *active_word, int64_t active_below) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(var_len > 0, "var_len must be larger than 0, but got ", var_len);
    TORCH_CHECK(out_len > 0, "out_len must be larger than 0, but got ", out_len);
    TORCH_CHECK(const_samples_len > 0, "const_samples_len must be larger than 0, but got ", const_samples_len);
    TORCH_CHECK(out_prob >= 0 && out_prob <= 1, "out_prob must be in range [0, 1], but got ", out_prob);
    TORCH_CHECK(const_prob >= 0 && const_prob <= 1, "const_prob must be in range [0, 1], but got ", const_prob);
    TORCH_CHECK(tree_index_offset >= 0 && tree_index_offset < (1LL << 32), "tree_index_offset must fit in 32 bits");
    TORCH_CHECK(active_below >= 0 && active_below < (1LL << 32), "active_below must fit in 32 bits");
    const c10::Device dev = keys_in.device();
    check_tensor(keys_in, {2}, "keys", dev);
    check_tensor(depth2leaf_probs, {EVOGP_MAX_FULL_DEPTH}, "depth2leaf_probs", dev, at::kFloat);
    check_tensor(roulette_funcs, {EVOGP_NUM_FUNCS}, "roulette_funcs", dev, at::kFloat);
    check_tensor(const_samples, {const_samples_len}, "const_samples", dev, at::kFloat);
    if (active_word) check_tensor(*active_word, {pop_size}, "active_word", dev, at::kInt);
    const Tensor keys = keys_u32(keys_in);
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(pop_size, gp_len, dev);
    const int rc = evogp_hip_generate_masked(
        (unsigned)pop_size, (unsigned)gp_len, (unsigned)var_len, (unsigned)out_len, (unsigned)const_samples_len, (float)out_prob,
        (float)const_prob, (const unsigned *)keys.data_ptr(), depth2leaf_probs.data_ptr<float>(), roulette_funcs.data_ptr<float>(),
        const_samples.data_ptr<float>(), std::get<0>(out).data_ptr<float>(), std::get<1>(out).data_ptr<int16_t>(),
        std::get<2>(out).data_ptr<int16_t>(), (unsigned)tree_index_offset, active_word ? active_word->data_ptr<int>() : nullptr,
        (unsigned)active_below, current_stream(dev));
    check_rc(rc, "tree_generate");
    return out;
}

// rows of trees whose counter-based word (4, n + tree_index_offset) of (seed, generation) is not below active_below are left
// uninitialised; no keys / active_word tensors (include/evogp_hip.h evogp_hip_generate_masked_hashed)
Tensor3 tree_generate_masked_hashed(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, int64_t const_samples_len, double out_prob,
                                    double const_prob, const Tensor &depth2leaf_probs, const Tensor &roulette_funcs, const Tensor &const_samples,
                                    int64_t tree_index_offset, int64_t seed, int64_t generation, int64_t active_below) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(var_len > 0 && out_len > 0 && const_samples_len > 0, "var_len, out_len and const_samples_len must be larger than 0");
    TORCH_CHECK(out_prob >= 0 && out_prob <= 1 && const_prob >= 0 && const_prob <= 1, "out_prob / const_prob must be in range [0, 1]");
    TORCH_CHECK(tree_index_offset >= 0 && tree_index_offset < (1LL << 32) && active_below >= 0 && active_below < (1LL << 32),
                "tree_index_offset / active_below must fit in 32 bits");
    const c10::Device dev = depth2leaf_probs.device();
    check_tensor(depth2leaf_probs, {EVOGP_MAX_FULL_DEPTH}, "depth2leaf_probs", dev, at::kFloat);
    check_tensor(roulette_funcs, {EVOGP_NUM_FUNCS}, "roulette_funcs", dev, at::kFloat);
    check_tensor(const_samples, {const_samples_len}, "const_samples", dev, at::kFloat);
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(pop_size, gp_len, dev);
    const int rc = evogp_hip_generate_masked_hashed(
        (unsigned)pop_size, (unsigned)gp_len, (unsigned)var_len, (unsigned)out_len, (unsigned)const_samples_len, (float)out_prob, (float)const_prob,
        depth2leaf_probs.data_ptr<float>(), roulette_funcs.data_ptr<float>(), const_samples.data_ptr<float>(), std::get<0>(out).data_ptr<float>(),
        std::get<1>(out).data_ptr<int16_t>(), std::get<2>(out).data_ptr<int16_t>(), (unsigned)tree_index_offset, seed, generation,
        (unsigned)active_below, current_stream(dev));
    check_rc(rc, "tree_generate_masked_hashed");
    return out;
}

// ---- the reference's five ops -------------------------------------------------------------------------------------------
Tensor3 tree_generate(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, int64_t const_samples_len, double out_prob,
                      double const_prob, const Tensor &keys, const Tensor &depth2leaf_probs, const Tensor &roulette_funcs,
                      const Tensor &const_samples) {
    return generate_impl(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys, depth2leaf_probs,
                         roulette_funcs, const_samples, 0, nullptr, 0);
}

Tensor3 tree_mutate(int64_t pop_size, int64_t gp_len, const Tensor &value_ori, const Tensor &type_ori, const Tensor &size_ori,
                    const Tensor &mutate_indices, const Tensor &value_new, const Tensor &type_new, const Tensor &size_new) {
    check_sizes(pop_size, gp_len);
    const c10::Device dev = value_new.device();
    check_forest(pop_size, gp_len, value_ori, type_ori, size_ori, dev, "_ori");
    check_tensor(mutate_indices, {pop_size}, "mutateIndices", dev, at::kInt);
    check_forest(pop_size, gp_len, value_new, type_new, size_new, dev, "_new");
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(pop_size, gp_len, dev);
    const int rc = evogp_hip_mutate((int)pop_size, (int)gp_len, value_ori.data_ptr<float>(), type_ori.data_ptr<int16_t>(),
                                    size_ori.data_ptr<int16_t>(), mutate_indices.data_ptr<int>(), value_new.data_ptr<float>(),
                                    type_new.data_ptr<int16_t>(), size_new.data_ptr<int16_t>(), std::get<0>(out).data_ptr<float>(),
                                    std::get<1>(out).data_ptr<int16_t>(), std::get<2>(out).data_ptr<int16_t>(), current_stream(dev));
    check_rc(rc, "tree_mutate");
    return out;
}

Tensor3 tree_crossover(int64_t pop_size_ori, int64_t pop_size_new, int64_t gp_len, const Tensor &value_ori, const Tensor &type_ori,
                       const Tensor &size_ori, const Tensor &left_idx, const Tensor &right_idx, const Tensor &left_node_idx,
                       const Tensor &right_node_idx) {
    TORCH_CHECK(pop_size_ori > 0, "pop_size_ori must be larger than 0, but got ", pop_size_ori);
    TORCH_CHECK(pop_size_new > 0, "pop_size_new must be larger than 0, but got ", pop_size_new);
    TORCH_CHECK(gp_len > 0 && gp_len <= kMaxStack, "gp_len must be in range (0, ", kMaxStack, "], but got ", gp_len);
    const c10::Device dev = value_ori.device();
    check_forest(pop_size_ori, gp_len, value_ori, type_ori, size_ori, dev, "_ori");
    check_tensor(left_idx, {pop_size_new}, "left_idx", dev, at::kInt);
    check_tensor(right_idx, {pop_size_new}, "right_idx", dev, at::kInt);
    check_tensor(left_node_idx, {pop_size_new}, "left_node_idx", dev, at::kInt);
    check_tensor(right_node_idx, {pop_size_new}, "right_node_idx", dev, at::kInt);
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(pop_size_new, gp_len, dev);
    const int rc = evogp_hip_crossover((int)pop_size_ori, (int)pop_size_new, (int)gp_len, value_ori.data_ptr<float>(),
                                       type_ori.data_ptr<int16_t>(), size_ori.data_ptr<int16_t>(), left_idx.data_ptr<int>(),
                                       right_idx.data_ptr<int>(), left_node_idx.data_ptr<int>(), right_node_idx.data_ptr<int>(),
                                       std::get<0>(out).data_ptr<float>(), std::get<1>(out).data_ptr<int16_t>(),
                                       std::get<2>(out).data_ptr<int16_t>(), current_stream(dev));
    check_rc(rc, "tree_crossover");
    return out;
}

Tensor tree_evaluate(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, const Tensor &value, const Tensor &type,
                     const Tensor &size, const Tensor &variables) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(var_len > 0, "var_len must be larger than 0, but got ", var_len);
    TORCH_CHECK(out_len > 0, "out_len must be larger than 0, but got ", out_len);
    const c10::Device dev = value.device();
    check_forest(pop_size, gp_len, value, type, size, dev);
    check_tensor(variables, {pop_size, var_len}, "variables", dev, at::kFloat);
    c10::DeviceGuard guard(dev);
    Tensor results = at::empty({pop_size, out_len}, value.options());
    const int rc = evogp_hip_evaluate((unsigned)pop_size, (unsigned)gp_len, (unsigned)var_len, (unsigned)out_len, value.data_ptr<float>(),
                                      type.data_ptr<int16_t>(), size.data_ptr<int16_t>(), variables.data_ptr<float>(),
                                      results.data_ptr<float>(), current_stream(dev));
    check_rc(rc, "tree_evaluate");
    return results;
}

Tensor sr_fitness_impl(int64_t pop_size, int64_t data_points, int64_t gp_len, int64_t var_len, int64_t out_len, bool use_mse,
                       const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &variables, const Tensor &labels,
                       int64_t kernel_type, int64_t func_mask = 0) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(var_len > 0, "var_len must be larger than 0, but got ", var_len);
    TORCH_CHECK(out_len > 0, "out_len must be larger than 0, but got ", out_len);
    TORCH_CHECK(data_points > 0, "data_points must be larger than 0, but got ", data_points);
    TORCH_CHECK(kernel_type >= 0 && kernel_type <= 4, "kernel_type must be in 0..4, but got ", kernel_type);
    const c10::Device dev = value.device();
    check_forest(pop_size, gp_len, value, type, size, dev);
    check_tensor(variables, {data_points, var_len}, "variables", dev, at::kFloat);
    check_tensor(labels, {data_points, out_len}, "labels", dev, at::kFloat);
    c10::DeviceGuard guard(dev);
    Tensor fitness = at::empty({pop_size}, value.options());
    TORCH_CHECK(func_mask >= 0 && func_mask < (1LL << 32), "func_mask must fit in 32 bits");
    const int rc = evogp_hip_sr_fitness_hinted((unsigned)pop_size, (unsigned)data_points, (unsigned)gp_len, (unsigned)var_len, (unsigned)out_len,
                                               use_mse ? 1 : 0, value.data_ptr<float>(), type.data_ptr<int16_t>(), size.data_ptr<int16_t>(),
                                               variables.data_ptr<float>(), labels.data_ptr<float>(), fitness.data_ptr<float>(),
                                               (unsigned)kernel_type, (unsigned)func_mask, current_stream(dev));
    check_rc(rc, "tree_SR_fitness");
    return fitness;
}

Tensor tree_SR_fitness(int64_t pop_size, int64_t data_points, int64_t gp_len, int64_t var_len, int64_t out_len, bool use_mse,
                       const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &variables, const Tensor &labels,
                       int64_t kernel_type) {
    return sr_fitness_impl(pop_size, data_points, gp_len, var_len, out_len, use_mse, value, type, size, variables, labels, kernel_type);
}

// tree_SR_fitness for a caller that knows the set of functions that can occur in the forest (0: unknown) -- include/evogp_hip.h
Tensor tree_SR_fitness_masked(int64_t pop_size, int64_t data_points, int64_t gp_len, int64_t var_len, int64_t out_len, bool use_mse,
                              const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &variables, const Tensor &labels,
                              int64_t kernel_type, int64_t func_mask) {
    return sr_fitness_impl(pop_size, data_points, gp_len, var_len, out_len, use_mse, value, type, size, variables, labels, kernel_type, func_mask);
}

// ---- extra ops (no counterpart in the reference) ------------------------------------------------------------------------
Tensor3 tree_generate_offset(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, int64_t const_samples_len,
                             double out_prob, double const_prob, const Tensor &keys, const Tensor &depth2leaf_probs,
                             const Tensor &roulette_funcs, const Tensor &const_samples, int64_t tree_index_offset) {
    return generate_impl(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys, depth2leaf_probs,
                         roulette_funcs, const_samples, tree_index_offset, nullptr, 0);
}

// rows of trees with active_word[n] >= active_below are left uninitialised
Tensor3 tree_generate_masked(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, int64_t const_samples_len,
                             double out_prob, double const_prob, const Tensor &keys, const Tensor &depth2leaf_probs,
                             const Tensor &roulette_funcs, const Tensor &const_samples, int64_t tree_index_offset,
                             const Tensor &active_word, int64_t active_below) {
    return generate_impl(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys, depth2leaf_probs,
                         roulette_funcs, const_samples, tree_index_offset, &active_word, active_below);
}

Tensor tree_batch_evaluate(int64_t pop_size, int64_t data_points, int64_t gp_len, int64_t var_len, int64_t out_len, const Tensor &value,
                           const Tensor &type, const Tensor &size, const Tensor &variables) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(var_len > 0 && out_len > 0 && data_points > 0, "var_len, out_len and data_points must be positive");
    const c10::Device dev = value.device();
    check_forest(pop_size, gp_len, value, type, size, dev);
    check_tensor(variables, {data_points, var_len}, "variables", dev, at::kFloat);
    c10::DeviceGuard guard(dev);
    Tensor results = at::empty({pop_size, data_points, out_len}, value.options());
    const int rc = evogp_hip_batch_evaluate((unsigned)pop_size, (unsigned)data_points, (unsigned)gp_len, (unsigned)var_len,
                                            (unsigned)out_len, value.data_ptr<float>(), type.data_ptr<int16_t>(), size.data_ptr<int16_t>(),
                                            variables.data_ptr<float>(), results.data_ptr<float>(), current_stream(dev));
    check_rc(rc, "tree_batch_evaluate");
    return results;
}

// counts[t] = #rows whose arg-max output (as torch.argmax(clip(softmax(.))) sees it) equals the int32 label
Tensor tree_batch_argmax_count(int64_t pop_size, int64_t data_points, int64_t gp_len, int64_t var_len, int64_t out_len,
                               const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &variables, const Tensor &labels) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(data_points > 0 && var_len > 0, "data_points and var_len must be larger than 0");
    TORCH_CHECK(out_len >= 2 && out_len <= 16, "out_len must be in [2, 16], but got ", out_len);
    const c10::Device dev = value.device();
    check_forest(pop_size, gp_len, value, type, size, dev);
    check_tensor(variables, {data_points, var_len}, "variables", dev, at::kFloat);
    check_tensor(labels, {data_points}, "labels", dev, at::kInt);
    c10::DeviceGuard guard(dev);
    Tensor counts = at::empty({pop_size}, at::TensorOptions().dtype(at::kInt).device(dev));
    const int rc = evogp_hip_batch_argmax_count((unsigned)pop_size, (unsigned)data_points, (unsigned)gp_len, (unsigned)var_len,
                                                (unsigned)out_len, value.data_ptr<float>(), type.data_ptr<int16_t>(),
                                                size.data_ptr<int16_t>(), variables.data_ptr<float>(), labels.data_ptr<int>(),
                                                (unsigned *)counts.data_ptr<int>(), current_stream(dev));
    check_rc(rc, "tree_batch_argmax_count");
    return counts;
}

// the operation lists of a multi-output forest (evaluate_prepared.hip): -> (workspace uint8[bytes], info int32[16]; info[0] = trees
// the lists cannot express, which tree_evaluate_prepared then has to send through the stack interpreter)
std::tuple<Tensor, Tensor> tree_evaluate_prepare(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, const Tensor &value,
                                                 const Tensor &type, const Tensor &size) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(var_len > 0 && var_len <= 255, "var_len must be in [1, 255], but got ", var_len);
    TORCH_CHECK(out_len >= 2 && out_len <= 32, "out_len must be in [2, 32], but got ", out_len);
    const c10::Device dev = value.device();
    check_forest(pop_size, gp_len, value, type, size, dev);
    c10::DeviceGuard guard(dev);
    const int64_t bytes = (int64_t)evogp_hip_evaluate_workspace_bytes((unsigned)pop_size, (unsigned)gp_len);
    Tensor ws = at::empty({bytes}, at::TensorOptions().dtype(at::kByte).device(dev));
    const int rc = evogp_hip_evaluate_prepare((unsigned)pop_size, (unsigned)gp_len, (unsigned)var_len, (unsigned)out_len, value.data_ptr<float>(),
                                              type.data_ptr<int16_t>(), size.data_ptr<int16_t>(), ws.data_ptr(), (size_t)bytes, current_stream(dev));
    check_rc(rc, "tree_evaluate_prepare");
    Tensor info = ws.narrow(0, bytes - 64, 64).view(at::kInt);
    return {ws, info};
}

Tensor tree_evaluate_prepared(int64_t pop_size, int64_t gp_len, int64_t var_len, int64_t out_len, const Tensor &value, const Tensor &type,
                              const Tensor &size, const Tensor &workspace, bool with_fallback, const Tensor &variables) {
    check_sizes(pop_size, gp_len);
    const c10::Device dev = value.device();
    check_forest(pop_size, gp_len, value, type, size, dev);
    check_tensor(variables, {pop_size, var_len}, "variables", dev, at::kFloat);
    check_tensor(workspace, {(int64_t)evogp_hip_evaluate_workspace_bytes((unsigned)pop_size, (unsigned)gp_len)}, "workspace", dev, at::kByte);
    c10::DeviceGuard guard(dev);
    Tensor results = at::empty({pop_size, out_len}, value.options());
    const int rc = evogp_hip_evaluate_prepared((unsigned)pop_size, (unsigned)gp_len, (unsigned)var_len, (unsigned)out_len, value.data_ptr<float>(),
                                               type.data_ptr<int16_t>(), size.data_ptr<int16_t>(), workspace.data_ptr(), with_fallback ? 1 : 0,
                                               variables.data_ptr<float>(), results.data_ptr<float>(), current_stream(dev));
    check_rc(rc, "tree_evaluate_prepared");
    return results;
}

// int32[rows][n_cols] whose columns [lo, hi) hold the counter-based random words of (seed, generation) (breed.hip); the other
// columns are uninitialised
// scores = NaN -> -inf, else -errors (negate) or errors: the sign of SymbolicRegression.evaluate and the NaN scrub of StandardPipeline.step in one launch
Tensor fitness_scores(const Tensor &errors, bool negate) {
    TORCH_CHECK(errors.is_cuda() && errors.scalar_type() == at::kFloat && errors.is_contiguous() && errors.dim() == 1 && errors.numel() > 0,
                "fitness_scores: errors must be a non-empty contiguous float32 CUDA vector");
    const c10::DeviceGuard guard(errors.device());
    Tensor out = at::empty_like(errors);
    check_rc(evogp_hip_fitness_scores((unsigned)errors.numel(), negate ? 1 : 0, errors.data_ptr<float>(), out.data_ptr<float>(), current_stream(errors.device())),
             "fitness_scores");
    return out;
}

Tensor random_words(int64_t seed, int64_t generation, int64_t rows, int64_t n_cols, int64_t lo, int64_t hi, c10::Device device) {
    TORCH_CHECK(device.is_cuda(), "random_words: the native generator runs on the GPU");
    TORCH_CHECK(rows > 0 && n_cols > 0 && lo >= 0 && lo <= hi && hi <= n_cols, "random_words: column range out of the array");
    c10::DeviceGuard guard(device);
    Tensor out = at::empty({rows, n_cols}, at::TensorOptions().dtype(at::kInt).device(device));
    check_rc(evogp_hip_random_words(seed, generation, (int)rows, n_cols, lo, hi, out.data_ptr<int>(), current_stream(out.device())), "random_words");
    return out;
}

// int32[n_keep]: the n_elite best trees, then the other survivors, each group in ascending index (select.hip)
Tensor select_survivors(const Tensor &fitness, int64_t n_elite, int64_t n_keep) {
    TORCH_CHECK(fitness.is_cuda() && fitness.is_contiguous() && fitness.scalar_type() == at::kFloat && fitness.dim() == 1,
                "fitness must be a contiguous float32 CUDA vector");
    const int64_t n = fitness.size(0);
    TORCH_CHECK(n > 0 && n_keep > 0 && n_keep <= n && n_elite >= 0 && n_elite <= n_keep, "need 0 <= n_elite <= n_keep <= n, got ", n_elite, ", ", n_keep, ", ", n);
    const c10::Device dev = fitness.device();
    c10::DeviceGuard guard(dev);
    Tensor order = at::empty({n_keep}, at::TensorOptions().dtype(at::kInt).device(dev));
    const int64_t words = (int64_t)(evogp_hip_select_workspace_bytes() / 4);
    const evogp_stream_t stream = current_stream(dev);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (capturing) {   // a replayed graph cannot alternate: a fresh zeroed workspace per captured call
        Tensor ws = at::zeros({words}, at::TensorOptions().dtype(at::kInt).device(dev));
        check_rc(evogp_hip_select((unsigned)n, (unsigned)n_elite, (unsigned)n_keep, fitness.data_ptr<float>(), order.data_ptr<int>(), ws.data_ptr(),
                                  stream), "select_survivors");
        return order;
    }
    // two workspaces per (device, stream): this call's was zeroed by the previous call's kernel on the same stream
    struct Slot { Tensor ws; int parity = 0; };
    static std::mutex mu;
    // (leaked on purpose: a static container of tensors would be destroyed at process teardown, after the caching allocator)
    static auto &slots = *new std::map<std::pair<int, void *>, Slot>();
    std::lock_guard<std::mutex> lock(mu);
    Slot &sl = slots[{(int)dev.index(), (void *)stream}];
    if (!sl.ws.defined()) sl.ws = at::zeros({2 * words}, at::TensorOptions().dtype(at::kInt).device(dev));
    int *base = sl.ws.data_ptr<int>();
    int *mine = base + (int64_t)sl.parity * words, *next = base + (int64_t)(1 - sl.parity) * words;
    const int rc = evogp_hip_select_alternating((unsigned)n, (unsigned)n_elite, (unsigned)n_keep, fitness.data_ptr<float>(), order.data_ptr<int>(), mine,
                                                next, stream);
    if (rc != EVOGP_OK) sl.ws = Tensor();   // the halves' zero / dirty protocol is unknown after a failed launch: a fresh workspace next time
    else sl.parity ^= 1;                    // (only a launch that ran has zeroed `next`)
    check_rc(rc, "select_survivors");
    return order;
}

// int32[n_tournaments]: the winner of every tournament of t_size counter-based contenders (select.hip)
Tensor tournament_select(const Tensor &fitness, int64_t n_tournaments, int64_t t_size, int64_t seed, int64_t generation) {
    TORCH_CHECK(fitness.is_cuda() && fitness.is_contiguous() && fitness.scalar_type() == at::kFloat && fitness.dim() == 1,
                "fitness must be a contiguous float32 CUDA vector");
    TORCH_CHECK(fitness.size(0) > 0 && n_tournaments > 0 && t_size > 0, "need a population, tournaments and contenders");
    const c10::Device dev = fitness.device();
    c10::DeviceGuard guard(dev);
    Tensor winners = at::empty({n_tournaments}, at::TensorOptions().dtype(at::kInt).device(dev));
    check_rc(evogp_hip_tournament_select((unsigned)fitness.size(0), (unsigned)n_tournaments, (unsigned)t_size, seed, generation,
                                         fitness.data_ptr<float>(), winners.data_ptr<int>(), current_stream(dev)), "tournament_select");
    return winners;
}

void check_order(const Tensor &order, int64_t need, const c10::Device &dev) {
    TORCH_CHECK(order.is_cuda() && order.is_contiguous() && order.scalar_type() == at::kInt && order.dim() == 1 && order.size(0) >= need &&
                    order.device() == dev,
                "order must be a contiguous int32 CUDA vector of >= max(n_elite, n_surv) entries on the forest's device");
}

std::tuple<Tensor, Tensor, Tensor, Tensor> breed_default(int64_t pop_size, int64_t gp_len, int64_t n_elite, int64_t n_surv, const Tensor &value,
                                                         const Tensor &type, const Tensor &size, const Tensor &order, const Tensor &rnd,
                                                         int64_t mutate_below, const Tensor &donor_value, const Tensor &donor_type,
                                                         const Tensor &donor_size, bool want_decisions) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(n_elite >= 0 && n_elite <= pop_size, "n_elite must be in [0, pop_size], but got ", n_elite);
    TORCH_CHECK(n_surv > 0 && n_surv <= pop_size, "n_surv must be in (0, pop_size], but got ", n_surv);
    TORCH_CHECK(mutate_below >= 0 && mutate_below < (1LL << 32), "mutate_below must fit in 32 bits");
    const c10::Device dev = value.device();
    check_forest(pop_size, gp_len, value, type, size, dev);
    const int64_t n_new = pop_size - n_elite;
    check_order(order, std::max(n_elite, n_surv), dev);
    check_tensor(rnd, {6, n_new}, "rnd", dev, at::kInt);
    check_forest(n_new, gp_len, donor_value, donor_type, donor_size, dev, " (donor)");
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(pop_size, gp_len, dev);
    Tensor dec = at::empty({want_decisions ? n_new : 0, 6}, at::TensorOptions().dtype(at::kInt).device(dev));
    const int rc = evogp_hip_breed_default((int)pop_size, (int)gp_len, (int)n_elite, (int)n_surv, value.data_ptr<float>(),
                                           type.data_ptr<int16_t>(), size.data_ptr<int16_t>(), order.data_ptr<int>(), rnd.data_ptr<int>(),
                                           (unsigned)mutate_below, donor_value.data_ptr<float>(), donor_type.data_ptr<int16_t>(),
                                           donor_size.data_ptr<int16_t>(), std::get<0>(out).data_ptr<float>(),
                                           std::get<1>(out).data_ptr<int16_t>(), std::get<2>(out).data_ptr<int16_t>(),
                                           want_decisions ? dec.data_ptr<int>() : nullptr, current_stream(dev));
    check_rc(rc, "breed_default");
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out), dec};
}

// Rows [row_begin, row_begin + row_count) of the next generation.  value / type / size: the whole population, or only the
// trees `order` names (a sharded run's survivor table).  The donor arrays are aligned with the OFFSPRING rows of the range:
// they may cover the whole range (row_count rows; the rows of elites are never read) or only its offspring (row_count minus
// the elite rows at the head of the range) -- then no padding copy is needed.
Tensor3 breed_default_rows(int64_t pop_size, int64_t gp_len, int64_t n_elite, int64_t n_surv, const Tensor &value, const Tensor &type,
                           const Tensor &size, const Tensor &order, const Tensor &rnd, int64_t mutate_below, const Tensor &donor_value,
                           const Tensor &donor_type, const Tensor &donor_size, int64_t row_begin, int64_t row_count) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(n_elite >= 0 && n_elite <= pop_size && n_surv > 0 && n_surv <= pop_size, "n_elite / n_surv out of range");
    TORCH_CHECK(row_begin >= 0 && row_count > 0 && row_begin + row_count <= pop_size, "row range out of the population");
    TORCH_CHECK(mutate_below >= 0 && mutate_below < (1LL << 32), "mutate_below must fit in 32 bits");
    TORCH_CHECK(value.dim() == 2 && value.size(0) > 0, "value must be a (rows, gp_len) tensor");
    const int64_t table_rows = value.size(0);
    const c10::Device dev = value.device();
    check_forest(table_rows, gp_len, value, type, size, dev);
    check_order(order, std::max(n_elite, n_surv), dev);
    check_tensor(rnd, {6, pop_size - n_elite}, "rnd", dev, at::kInt);
    const int64_t head = std::max<int64_t>(0, std::min(row_begin + row_count, n_elite) - row_begin);  // elite rows at the head of the range
    const int64_t drows = donor_value.dim() == 2 ? donor_value.size(0) : -1;
    TORCH_CHECK(drows == row_count || drows == row_count - head, "donor arrays must have ", row_count, " or ", row_count - head,
                " rows, but got ", drows);
    check_forest(drows, gp_len, donor_value, donor_type, donor_size, dev, " (donor)");
    const int64_t skip = drows == row_count - head ? head : 0;  // the engine indexes donors by (row - row_begin)
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(row_count, gp_len, dev);
    const int rc = evogp_hip_breed_default_table(
        (int)pop_size, (int)table_rows, (int)gp_len, (int)n_elite, (int)n_surv, value.data_ptr<float>(), type.data_ptr<int16_t>(),
        size.data_ptr<int16_t>(), order.data_ptr<int>(), rnd.data_ptr<int>(), (unsigned)mutate_below,
        donor_value.data_ptr<float>() - skip * gp_len, donor_type.data_ptr<int16_t>() - skip * gp_len,
        donor_size.data_ptr<int16_t>() - skip * gp_len, std::get<0>(out).data_ptr<float>(), std::get<1>(out).data_ptr<int16_t>(),
        std::get<2>(out).data_ptr<int16_t>(), nullptr, (int)row_begin, (int)row_count, current_stream(dev));
    check_rc(rc, "breed_default_rows");
    return out;
}

// The same rows for ANY selection operator: the elites and the parents are two lists of table rows (parents may repeat, as
// the survivor indices of a tournament selection do); n_elite / n_surv are the lists' lengths.
Tensor3 breed_rows_impl(int64_t pop_size, int64_t gp_len, const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &elite_rows,
                        const Tensor &parent_rows, const Tensor *rnd_or_null, int64_t seed, int64_t generation, int64_t mutate_below,
                        const Tensor &donor_value, const Tensor &donor_type, const Tensor &donor_size, int64_t row_begin, int64_t row_count) {
    check_sizes(pop_size, gp_len);
    TORCH_CHECK(row_begin >= 0 && row_count > 0 && row_begin + row_count <= pop_size, "row range out of the population");
    TORCH_CHECK(mutate_below >= 0 && mutate_below < (1LL << 32), "mutate_below must fit in 32 bits");
    TORCH_CHECK(value.dim() == 2 && value.size(0) > 0, "value must be a (rows, gp_len) tensor");
    const int64_t table_rows = value.size(0);
    const c10::Device dev = value.device();
    check_forest(table_rows, gp_len, value, type, size, dev);
    TORCH_CHECK(elite_rows.dim() == 1 && parent_rows.dim() == 1, "elite_rows / parent_rows must be vectors");
    const int64_t n_elite = elite_rows.size(0), n_surv = parent_rows.size(0);
    TORCH_CHECK(n_elite <= pop_size && n_surv > 0, "need n_elite <= pop_size and at least one parent, got ", n_elite, ", ", n_surv);
    check_order(parent_rows, n_surv, dev);
    if (n_elite > 0) check_order(elite_rows, n_elite, dev);
    if (rnd_or_null) check_tensor(*rnd_or_null, {6, pop_size - n_elite}, "rnd", dev, at::kInt);
    const int64_t head = std::max<int64_t>(0, std::min(row_begin + row_count, n_elite) - row_begin);
    const int64_t drows = donor_value.dim() == 2 ? donor_value.size(0) : -1;
    TORCH_CHECK(drows == row_count || drows == row_count - head, "donor arrays must have ", row_count, " or ", row_count - head,
                " rows, but got ", drows);
    check_forest(drows, gp_len, donor_value, donor_type, donor_size, dev, " (donor)");
    const int64_t skip = drows == row_count - head ? head : 0;
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(row_count, gp_len, dev);
    const int rc = rnd_or_null
        ? evogp_hip_breed_lists(
              (int)pop_size, (int)table_rows, (int)gp_len, (int)n_elite, (int)n_surv, value.data_ptr<float>(), type.data_ptr<int16_t>(),
              size.data_ptr<int16_t>(), n_elite > 0 ? elite_rows.data_ptr<int>() : nullptr, parent_rows.data_ptr<int>(), rnd_or_null->data_ptr<int>(),
              (unsigned)mutate_below, donor_value.data_ptr<float>() - skip * gp_len, donor_type.data_ptr<int16_t>() - skip * gp_len,
              donor_size.data_ptr<int16_t>() - skip * gp_len, std::get<0>(out).data_ptr<float>(), std::get<1>(out).data_ptr<int16_t>(),
              std::get<2>(out).data_ptr<int16_t>(), nullptr, (int)row_begin, (int)row_count, current_stream(dev))
        : evogp_hip_breed_lists_hashed(
              (int)pop_size, (int)table_rows, (int)gp_len, (int)n_elite, (int)n_surv, value.data_ptr<float>(), type.data_ptr<int16_t>(),
              size.data_ptr<int16_t>(), n_elite > 0 ? elite_rows.data_ptr<int>() : nullptr, parent_rows.data_ptr<int>(), seed, generation,
              (unsigned)mutate_below, donor_value.data_ptr<float>() - skip * gp_len, donor_type.data_ptr<int16_t>() - skip * gp_len,
              donor_size.data_ptr<int16_t>() - skip * gp_len, std::get<0>(out).data_ptr<float>(), std::get<1>(out).data_ptr<int16_t>(),
              std::get<2>(out).data_ptr<int16_t>(), nullptr, (int)row_begin, (int)row_count, current_stream(dev));
    check_rc(rc, "breed_rows");
    return out;
}

Tensor3 breed_rows(int64_t pop_size, int64_t gp_len, const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &elite_rows,
                   const Tensor &parent_rows, const Tensor &rnd, int64_t mutate_below, const Tensor &donor_value, const Tensor &donor_type,
                   const Tensor &donor_size, int64_t row_begin, int64_t row_count) {
    return breed_rows_impl(pop_size, gp_len, value, type, size, elite_rows, parent_rows, &rnd, 0, 0, mutate_below, donor_value, donor_type, donor_size,
                           row_begin, row_count);
}

// breed_rows with the six words of every offspring computed in the kernel from (seed, generation) instead of read from `rnd`
Tensor3 breed_rows_hashed(int64_t pop_size, int64_t gp_len, const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &elite_rows,
                          const Tensor &parent_rows, int64_t seed, int64_t generation, int64_t mutate_below, const Tensor &donor_value,
                          const Tensor &donor_type, const Tensor &donor_size, int64_t row_begin, int64_t row_count) {
    return breed_rows_impl(pop_size, gp_len, value, type, size, elite_rows, parent_rows, nullptr, seed, generation, mutate_below, donor_value,
                           donor_type, donor_size, row_begin, row_count);
}

// DeleteMutation (mode 0) / HoistMutation (mode 1) drawn and applied in one launch (include/evogp_hip.h evogp_hip_structural_mutate)
std::tuple<Tensor, Tensor, Tensor, Tensor> structural_mutate(int64_t mode, double rate, int64_t max_size, bool inner_is_offset, int64_t skip_rows, int64_t seed,
                                                             int64_t call, const Tensor &value, const Tensor &type, const Tensor &size, bool want_decisions) {
    TORCH_CHECK(value.dim() == 2, "value must be a (pop, gp_len) tensor");
    const int64_t pop = value.size(0), gp_len = value.size(1);
    check_sizes(pop, gp_len);
    const c10::Device dev = value.device();
    check_forest(pop, gp_len, value, type, size, dev);
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(pop, gp_len, dev);
    Tensor dec = want_decisions ? at::empty({pop, 2}, value.options().dtype(at::kInt)) : at::empty({0}, value.options().dtype(at::kInt));
    check_rc(evogp_hip_structural_mutate((int)pop, (int)gp_len, (int)mode, (float)rate, (int)max_size, inner_is_offset ? 1 : 0, (int)skip_rows, seed, call,
                                         value.data_ptr<float>(), type.data_ptr<int16_t>(), size.data_ptr<int16_t>(), std::get<0>(out).data_ptr<float>(),
                                         std::get<1>(out).data_ptr<int16_t>(), std::get<2>(out).data_ptr<int16_t>(),
                                         want_decisions ? dec.data_ptr<int>() : nullptr, current_stream(dev)),
             "structural_mutate");
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out), dec};
}

// InsertMutation drawn and applied in one launch over fresh trees from tree_generate_masked_hashed(seed, call, mutate_below) (evogp_hip_insert_mutate)
std::tuple<Tensor, Tensor, Tensor, Tensor> insert_mutate(int64_t mutate_below, int64_t skip_rows, int64_t seed, int64_t call, const Tensor &value,
                                                         const Tensor &type, const Tensor &size, const Tensor &donor_value, const Tensor &donor_type,
                                                         const Tensor &donor_size, bool want_decisions) {
    TORCH_CHECK(value.dim() == 2, "value must be a (pop, gp_len) tensor");
    const int64_t pop = value.size(0), gp_len = value.size(1);
    check_sizes(pop, gp_len);
    TORCH_CHECK(mutate_below >= 0 && mutate_below < (1LL << 32), "mutate_below must fit in 32 bits");
    const c10::Device dev = value.device();
    check_forest(pop, gp_len, value, type, size, dev);
    check_forest(pop, gp_len, donor_value, donor_type, donor_size, dev);
    c10::DeviceGuard guard(dev);
    Tensor3 out = empty_forest(pop, gp_len, dev);
    Tensor dec = want_decisions ? at::empty({pop, 2}, value.options().dtype(at::kInt)) : at::empty({0}, value.options().dtype(at::kInt));
    check_rc(evogp_hip_insert_mutate((int)pop, (int)gp_len, (unsigned)mutate_below, (int)skip_rows, seed, call, value.data_ptr<float>(),
                                     type.data_ptr<int16_t>(), size.data_ptr<int16_t>(), donor_value.data_ptr<float>(), donor_type.data_ptr<int16_t>(),
                                     donor_size.data_ptr<int16_t>(), std::get<0>(out).data_ptr<float>(), std::get<1>(out).data_ptr<int16_t>(),
                                     std::get<2>(out).data_ptr<int16_t>(), want_decisions ? dec.data_ptr<int>() : nullptr, current_stream(dev)),
             "insert_mutate");
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out), dec};
}

// Multi / Single Point / Const mutation drawn and applied in one launch: the new value array (evogp_hip_point_mutate)
Tensor point_mutate(int64_t mode, double rate, double intensity, bool per_node, bool modify_output, bool fix_roulette, int64_t skip_rows, int64_t input_len,
                    int64_t output_len, int64_t seed, int64_t call, const Tensor &value, const Tensor &type, const Tensor &size, const Tensor &roulette_ufuncs,
                    const Tensor &roulette_bfuncs, const Tensor &roulette_tfuncs, const Tensor &const_samples) {
    TORCH_CHECK(value.dim() == 2, "value must be a (pop, gp_len) tensor");
    const int64_t pop = value.size(0), gp_len = value.size(1);
    check_sizes(pop, gp_len);
    const c10::Device dev = value.device();
    check_forest(pop, gp_len, value, type, size, dev);
    check_tensor(roulette_ufuncs, {29}, "roulette_ufuncs", dev, at::kFloat);
    check_tensor(roulette_bfuncs, {29}, "roulette_bfuncs", dev, at::kFloat);
    check_tensor(roulette_tfuncs, {29}, "roulette_tfuncs", dev, at::kFloat);
    TORCH_CHECK(const_samples.dim() == 1 && const_samples.size(0) > 0, "const_samples must be a non-empty vector");
    check_tensor(const_samples, {const_samples.size(0)}, "const_samples", dev, at::kFloat);
    c10::DeviceGuard guard(dev);
    Tensor out = at::empty_like(value);
    check_rc(evogp_hip_point_mutate((int)pop, (int)gp_len, (int)mode, (float)rate, (float)intensity, per_node ? 1 : 0, modify_output ? 1 : 0, fix_roulette ? 1 : 0,
                                    (int)skip_rows, (int)input_len, (int)output_len, (int)const_samples.size(0), seed, call, value.data_ptr<float>(),
                                    type.data_ptr<int16_t>(), size.data_ptr<int16_t>(), roulette_ufuncs.data_ptr<float>(), roulette_bfuncs.data_ptr<float>(),
                                    roulette_tfuncs.data_ptr<float>(), const_samples.data_ptr<float>(), out.data_ptr<float>(), current_stream(dev)),
             "point_mutate");
    return out;
}

}  // namespace

// schemas of the reference, verbatim (torch_wrapper.cu:294-298)
TORCH_LIBRARY(evogp_cuda, m) {
    m.def("tree_generate(int i1, int i2, int i3, int i4, int i5, float f1, float f2, Tensor t1, Tensor t2, Tensor t3, Tensor t4) -> (Tensor t5, Tensor t6, Tensor t7)");
    m.def("tree_mutate(int i1, int i2, Tensor t1, Tensor t2, Tensor t3, Tensor t4, Tensor t5, Tensor t6, Tensor t7) -> (Tensor t8, Tensor t9, Tensor t10)");
    m.def("tree_crossover(int i1, int i2, int i3, Tensor t1, Tensor t2, Tensor t3, Tensor t4, Tensor t5, Tensor t6, Tensor t7) -> (Tensor t8, Tensor t9, Tensor t10)");
    m.def("tree_evaluate(int i1, int i2, int i3, int i4, Tensor t1, Tensor t2, Tensor t3, Tensor t4) -> Tensor t5");
    m.def("tree_SR_fitness(int i1, int i2, int i3, int i4, int i5, bool b1, Tensor t1, Tensor t2, Tensor t3, Tensor t4, Tensor t5, int i6) -> Tensor t6");
}

TORCH_LIBRARY_IMPL(evogp_cuda, CUDA, m) {
    m.impl("tree_generate", &tree_generate);
    m.impl("tree_mutate", &tree_mutate);
    m.impl("tree_crossover", &tree_crossover);
    m.impl("tree_evaluate", &tree_evaluate);
    m.impl("tree_SR_fitness", &tree_SR_fitness);
}

// the engine's record buffers come out of torch's caching allocator: torch.cuda.memory_allocated() / memory_summary() show them
// (include/evogp_hip.h evogp_hip_set_allocator; EVOGP_TORCH_ALLOCATOR=0: plain hipMalloc as in rounds 1-4)
static void *torch_pool_alloc(size_t bytes) {
    try {
        // Not while ANY stream of this thread's device is being captured: raw_alloc may then hand out memory of the
        // capture's private pool, and the device-wide wait the engine makes before it first touches a block would invalidate the
        // capture.  nullptr: the engine takes the call on kernels that need no record buffer.  (Blocks are returned to the pool only
        // by evogp_hip_release_workspaces, after a device-wide wait.)
        if (c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None) return nullptr;
        return c10::hip::HIPCachingAllocator::raw_alloc(bytes);
    } catch (...) {
        return nullptr;
    }
}
static void torch_pool_free(void *ptr) { c10::hip::HIPCachingAllocator::raw_delete(ptr); }
static const int installed_allocator = [] {
    const char *e = getenv("EVOGP_TORCH_ALLOCATOR");
    return (e && e[0] == '0') ? 0 : evogp_hip_set_allocator(&torch_pool_alloc, &torch_pool_free);
}();

TORCH_LIBRARY(evogp_hip, m) {
    m.def("tree_generate_offset(int pop_size, int gp_len, int var_len, int out_len, int const_samples_len, float out_prob, float const_prob,"
          " Tensor keys, Tensor depth2leaf_probs, Tensor roulette_funcs, Tensor const_samples, int tree_index_offset)"
          " -> (Tensor value, Tensor node_type, Tensor subtree_size)");
    m.def("tree_generate_masked(int pop_size, int gp_len, int var_len, int out_len, int const_samples_len, float out_prob, float const_prob,"
          " Tensor keys, Tensor depth2leaf_probs, Tensor roulette_funcs, Tensor const_samples, int tree_index_offset,"
          " Tensor active_word, int active_below) -> (Tensor value, Tensor node_type, Tensor subtree_size)");
    m.def("tree_batch_evaluate(int pop_size, int data_points, int gp_len, int var_len, int out_len, Tensor value, Tensor node_type,"
          " Tensor subtree_size, Tensor variables) -> Tensor results");
    m.def("tree_batch_argmax_count(int pop_size, int data_points, int gp_len, int var_len, int out_len, Tensor value, Tensor node_type,"
          " Tensor subtree_size, Tensor variables, Tensor labels) -> Tensor counts");
    m.def("tree_evaluate_prepare(int pop_size, int gp_len, int var_len, int out_len, Tensor value, Tensor node_type, Tensor subtree_size)"
          " -> (Tensor workspace, Tensor info)");
    m.def("tree_evaluate_prepared(int pop_size, int gp_len, int var_len, int out_len, Tensor value, Tensor node_type, Tensor subtree_size,"
          " Tensor workspace, bool with_fallback, Tensor variables) -> Tensor results");
    m.def("random_words(int seed, int generation, int rows, int n_cols, int lo, int hi, Device device) -> Tensor");
    m.def("fitness_scores(Tensor errors, bool negate) -> Tensor");
    m.def("select_survivors(Tensor fitness, int n_elite, int n_keep) -> Tensor");
    m.def("tournament_select(Tensor fitness, int n_tournaments, int t_size, int seed, int generation) -> Tensor");
    m.def("breed_default(int pop_size, int gp_len, int n_elite, int n_surv, Tensor value, Tensor node_type, Tensor subtree_size,"
          " Tensor order, Tensor rnd, int mutate_below, Tensor donor_value, Tensor donor_type, Tensor donor_size,"
          " bool want_decisions) -> (Tensor value, Tensor node_type, Tensor subtree_size, Tensor decisions)");
    m.def("breed_default_rows(int pop_size, int gp_len, int n_elite, int n_surv, Tensor value, Tensor node_type, Tensor subtree_size,"
          " Tensor order, Tensor rnd, int mutate_below, Tensor donor_value, Tensor donor_type, Tensor donor_size,"
          " int row_begin, int row_count) -> (Tensor value, Tensor node_type, Tensor subtree_size)");
    m.def("breed_rows(int pop_size, int gp_len, Tensor value, Tensor node_type, Tensor subtree_size, Tensor elite_rows, Tensor parent_rows,"
          " Tensor rnd, int mutate_below, Tensor donor_value, Tensor donor_type, Tensor donor_size, int row_begin, int row_count)"
          " -> (Tensor value, Tensor node_type, Tensor subtree_size)");
    m.def("breed_rows_hashed(int pop_size, int gp_len, Tensor value, Tensor node_type, Tensor subtree_size, Tensor elite_rows, Tensor parent_rows,"
          " int seed, int generation, int mutate_below, Tensor donor_value, Tensor donor_type, Tensor donor_size, int row_begin, int row_count)"
          " -> (Tensor value, Tensor node_type, Tensor subtree_size)");
    m.def("structural_mutate(int mode, float rate, int max_size, bool inner_is_offset, int skip_rows, int seed, int call, Tensor value, Tensor node_type,"
          " Tensor subtree_size, bool want_decisions) -> (Tensor value, Tensor node_type, Tensor subtree_size, Tensor decisions)");
    m.def("insert_mutate(int mutate_below, int skip_rows, int seed, int call, Tensor value, Tensor node_type, Tensor subtree_size, Tensor donor_value,"
          " Tensor donor_type, Tensor donor_size, bool want_decisions) -> (Tensor value, Tensor node_type, Tensor subtree_size, Tensor decisions)");
    m.def("point_mutate(int mode, float rate, float intensity, bool per_node, bool modify_output, bool fix_roulette, int skip_rows, int input_len, int output_len,"
          " int seed, int call, Tensor value, Tensor node_type, Tensor subtree_size, Tensor roulette_ufuncs, Tensor roulette_bfuncs, Tensor roulette_tfuncs,"
          " Tensor const_samples) -> Tensor");
    m.def("tree_generate_masked_hashed(int pop_size, int gp_len, int var_len, int out_len, int const_samples_len, float out_prob, float const_prob,"
          " Tensor depth2leaf_probs, Tensor roulette_funcs, Tensor const_samples, int tree_index_offset, int seed, int generation, int active_below)"
          " -> (Tensor value, Tensor node_type, Tensor subtree_size)");
    m.def("tree_SR_fitness_masked(int pop_size, int data_points, int gp_len, int var_len, int out_len, bool use_mse, Tensor value,"
          " Tensor node_type, Tensor subtree_size, Tensor variables, Tensor labels, int kernel_type, int func_mask) -> Tensor");
}

TORCH_LIBRARY_IMPL(evogp_hip, CompositeExplicitAutograd, m) { m.impl("random_words", &random_words); }  // no tensor argument to dispatch on

TORCH_LIBRARY_IMPL(evogp_hip, CUDA, m) {
    m.impl("tree_generate_offset", &tree_generate_offset);
    m.impl("tree_generate_masked", &tree_generate_masked);
    m.impl("tree_batch_evaluate", &tree_batch_evaluate);
    m.impl("tree_batch_argmax_count", &tree_batch_argmax_count);
    m.impl("tree_evaluate_prepare", &tree_evaluate_prepare);
    m.impl("tree_evaluate_prepared", &tree_evaluate_prepared);
    m.impl("breed_default", &breed_default);
    m.impl("breed_default_rows", &breed_default_rows);
    m.impl("breed_rows", &breed_rows);
    m.impl("breed_rows_hashed", &breed_rows_hashed);
    m.impl("tree_generate_masked_hashed", &tree_generate_masked_hashed);
    m.impl("structural_mutate", &structural_mutate);
    m.impl("insert_mutate", &insert_mutate);
    m.impl("point_mutate", &point_mutate);
    m.impl("tree_SR_fitness_masked", &tree_SR_fitness_masked);
    m.impl("select_survivors", &select_survivors);
    m.impl("fitness_scores", &fitness_scores);
    m.impl("tournament_select", &tournament_select);
}
