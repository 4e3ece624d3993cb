// az_cnn.hpp -- K3 interface: fused Connect4 policy/value CNN forward on MFMA.  Three kernels, one contract (1e-4 against
// the reference's fp32 forward; golden vectors from all four reference checkpoints), selected with az_net_set_kernel:
//   az_cnn_split.hip: the product kernel: direct convolution on fp16 MFMA, every fp32 product as three fp16 products on
//                     hi/lo split operands, fp32 accumulation, per-position dynamic range (AZ_K3_SPLIT)
//   az_cnn_wino.hip : the fp32-MFMA kernel (3x3 convolutions as Winograd F(2x2,3x3) products; AZ_K3_WINO)
//   az_cnn.hip      : dispatch + the direct implicit-GEMM fp32-MFMA kernel (A/B baseline; AZ_K3_DIRECT)
#pragma once
#include <math.h>

#include "az_internal.hpp"

namespace az {

struct Cnn;   // device-resident folded weights of the selected kernel
struct CnnW;  // Winograd-form weights
struct CnnS;  // split-fp16 form

// kind: AZ_K3_* of include/azamd.h (AZ_K3_DEFAULT: environment override, else the product kernel AZ_K3_PRODUCT)
constexpr int AZ_K3_PRODUCT = AZ_K3_SPLIT;
int cnn_create(const float* blob, size_t n_floats, int kind, Cnn** out);
int cnn_kind(const Cnn* c);
void cnn_destroy(Cnn* c);
// n packed Connect4 positions (device) -> optional logp[n][7], logv[n][3], probs[n][7], value[n] (device).
int cnn_forward(Cnn* c, i64 n, const u64* states_dev, float* logp, float* logv, float* probs, float* value,
                hipStream_t st);
// evaluate the compacted leaf list of a split search round: reads d.eval_list/eval_count/leaf_state,
// writes d.eval_probs/eval_value for those slots.
int cnn_eval_leaves(Cnn* c, const Dev& d, hipStream_t st);

int cnnw_create(const float* blob, size_t n_floats, CnnW** out);
void cnnw_destroy(CnnW* c);
// max_n bounds the batch (grid size); the batch size itself is n_host, or *n_ptr (device) when n_ptr != null.
int cnnw_launch(CnnW* c, unsigned max_n, i64 n_host, const u32* n_ptr, const u32* list, const u64* states, float* logp,
                float* logv, float* probs, float* value, hipStream_t st);

int cnns_create(const float* blob, size_t n_floats, CnnS** out);
void cnns_destroy(CnnS* c);
// grid_cap != 0: the launch may use a grid of that many workgroups that stride over the batch (capacity far above the usual fill)
int cnns_launch(CnnS* c, unsigned max_n, i64 n_host, const u32* n_ptr, const u32* list, const u64* states, float* logp,
                float* logv, float* probs, float* value, hipStream_t st, unsigned grid_cap = 0);

// ---- reference state_dict blob walking + eval-mode BatchNorm folding (Network.py:19-24,33-51) ----
constexpr float CNN_BN_EPS = 1e-5f;  // torch.nn.BatchNorm2d default
struct CnnBlobReader {
    const float* p;
    const float* take(size_t n) {
        const float* r = p;
        p += n;
        return r;
    }
};
struct CnnBn {
    const float *g, *b, *m, *v;
};
inline CnnBn cnn_take_bn(CnnBlobReader& br, size_t c) {
    CnnBn bn;
    bn.g = br.take(c);
    bn.b = br.take(c);
    bn.m = br.take(c);
    bn.v = br.take(c);
    return bn;
}
inline float cnn_bn_scale(const CnnBn& bn, int c) { return bn.g[c] / sqrtf(bn.v[c] + CNN_BN_EPS); }
inline float cnn_bn_bias(const CnnBn& bn, int c) { return bn.b[c] - bn.m[c] * cnn_bn_scale(bn, c); }

}  // namespace az
