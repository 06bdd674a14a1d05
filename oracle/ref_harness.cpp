// ref_harness.cpp — drives the REFERENCE's own device functions on the host CPU.
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: build_ref.py extracts the
// device-only line ranges of /root/reference/src/evogp/cuda/{forward,generate,mutation}.cu into a
// temporary directory at build time (never into the repo) and this harness #includes them there.
// The qualifier shim (cuda_runtime.h in that temporary directory) turns __global__/__device__ into
// nothing and blockIdx/blockDim/threadIdx into plain global structs, so a "launch" is a loop that
// sets g_blockIdx.x = n and calls the kernel as an ordinary function.
// The RNG is the real rocThrust taus88 — the same third-party engine the reference uses.
//
// Exposes the same five entry points as oracle/evogp_oracle.h with a ref_ prefix.
#include "kernel.h" // the reference's header, copied to the temp dir; pulls in the shim + rocThrust

#include "ref_forward_a.inc"  // forward.cu:79-351   _process_node, _treeGPEvalByStack, treeGPEvalKernel
#include "ref_forward_b.inc"  // forward.cu:373-400  calculate_fit
#include "ref_generate.inc"   // generate.cu:16-173  treeGPGenerate
#include "ref_mutation_a.inc" // mutation.cu:5-184   _gpTreeReplace, treeGPMutationKernel
#include "ref_mutation_b.inc" // mutation.cu:224-309 treeGPCrossoverKernel

#include <vector>

static void set_thread(unsigned n) {
    g_blockIdx.x = n; g_blockIdx.y = 0; g_blockDim.x = 1; g_threadIdx.x = 0;
}

// The SR kernels use __syncthreads/shared memory and cannot be emulated as a serial "launch";
// per SURVEY.md Appendix C step 3 the per-(tree, datapoint) device functions are called directly
// and the block reduction (forward.cu:456-471: pairwise over 1024 lanes, one partial per block,
// then /D at :474-479) is done here on the host in the same order.
template <bool MO, bool MSE>
static void sr_impl(unsigned pop, unsigned D, unsigned gp_len, unsigned var_len, unsigned out_len,
                    const float *v, const int16_t *t, const int16_t *s, const float *X, const float *y, float *fit) {
    std::vector<float> stack(MAX_STACK + 8);
    std::vector<int16_t> infos(2 * MAX_STACK + 8);
    std::vector<float> lane(1024);
    for (unsigned n = 0; n < pop; ++n) {
        float total = 0.f;
        for (unsigned base = 0; base < D; base += 1024) {
            for (unsigned l = 0; l < 1024; ++l) {
                unsigned d = base + l;
                float err = 0.f;
                if (d < D) {
                    float *s_outs = nullptr; int top = 0;
                    _treeGPEvalByStack<MO>(v + (size_t)n * gp_len, t + (size_t)n * gp_len, s + (size_t)n * gp_len,
                                           X + (size_t)d * var_len, stack.data(), infos.data(), pop, gp_len, var_len, out_len, s_outs, top);
                    err = calculate_fit<MO, MSE>(top, stack.data(), s_outs, y + (size_t)d * out_len, out_len);
                }
                lane[l] = err;
            }
            for (unsigned w = 512; w > 0; w >>= 1)
                for (unsigned l = 0; l < w; ++l) lane[l] += lane[l + w];
            total += lane[0];
        }
        fit[n] = total / D;
    }
}

extern "C" {

void ref_generate(unsigned pop, unsigned gp_len, unsigned var_len, unsigned out_len, unsigned n_const,
                  float out_prob, float const_prob, const unsigned *keys, const float *leaf,
                  const float *roulette, const float *consts, float *v, int16_t *t, int16_t *s) {
    for (unsigned n = 0; n < pop; ++n) {
        set_thread(n);
        if (out_len > 1)
            treeGPGenerate<true>(pop, gp_len, var_len, out_len, n_const, out_prob, const_prob, v, t, s, keys, leaf, roulette, consts);
        else
            treeGPGenerate<false>(pop, gp_len, var_len, out_len, n_const, out_prob, const_prob, v, t, s, keys, leaf, roulette, consts);
    }
}

void ref_mutate(int pop, int gp_len, const float *ov, const int16_t *ot, const int16_t *os, const int *idx,
                const float *nv, const int16_t *nt, const int16_t *ns, float *rv, int16_t *rt, int16_t *rs) {
    for (int n = 0; n < pop; ++n) {
        set_thread((unsigned)n);
        treeGPMutationKernel(ov, ot, os, idx, nv, nt, ns, rv, rt, rs, pop, gp_len);
    }
}

void ref_crossover(int pop_ori, int pop_new, int gp_len, const float *v, const int16_t *t, const int16_t *s,
                   const int *li, const int *ri, const int *ln, const int *rn, float *rv, int16_t *rt, int16_t *rs) {
    for (int n = 0; n < pop_new; ++n) {
        set_thread((unsigned)n);
        treeGPCrossoverKernel(pop_ori, pop_new, gp_len, v, t, s, li, ri, ln, rn, rv, rt, rs);
    }
}

void ref_evaluate(unsigned pop, unsigned gp_len, unsigned var_len, unsigned out_len, const float *v,
                  const int16_t *t, const int16_t *s, const float *vars, float *res) {
    for (unsigned n = 0; n < pop; ++n) {
        set_thread(n);
        if (out_len > 1) treeGPEvalKernel<true>(pop, gp_len, var_len, out_len, v, t, s, vars, res);
        else treeGPEvalKernel<false>(pop, gp_len, var_len, 1, v, t, s, vars, res);
    }
}

void ref_sr_fitness(unsigned pop, unsigned D, unsigned gp_len, unsigned var_len, unsigned out_len, int use_mse,
                    const float *v, const int16_t *t, const int16_t *s, const float *X, const float *y, float *fit) {
    if (out_len > 1) {
        if (use_mse) sr_impl<true, true>(pop, D, gp_len, var_len, out_len, v, t, s, X, y, fit);
        else sr_impl<true, false>(pop, D, gp_len, var_len, out_len, v, t, s, X, y, fit);
    } else {
        if (use_mse) sr_impl<false, true>(pop, D, gp_len, var_len, 1, v, t, s, X, y, fit);
        else sr_impl<false, false>(pop, D, gp_len, var_len, 1, v, t, s, X, y, fit);
    }
}

unsigned ref_hash(unsigned n, unsigned k1, unsigned k2) { return hash(n, k1, k2); }

void ref_taus88(unsigned seed, int count, unsigned *out, float *fout) {
    RandomEngine e(seed), e2(seed);
    thrust::uniform_real_distribution<float> rand(0.0f, 1.0f);
    for (int i = 0; i < count; ++i) {
        if (out) out[i] = e();
        if (fout) fout[i] = rand(e2);
    }
}

} // extern "C"
