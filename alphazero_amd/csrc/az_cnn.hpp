// az_cnn.hpp -- K3 interface: fused Connect4 policy/value CNN forward on fp32 MFMA (az_cnn.hip).
#pragma once
#include "az_internal.hpp"

namespace az {

struct Cnn;  // device-resident folded weights + scratch

int cnn_create(const float* blob, size_t n_floats, int max_batch, Cnn** out);
void cnn_destroy(Cnn* c);
// n packed Connect4 positions (device) -> optional logp[n][7], logv[n][3], probs[n][7], value[n] (device).
int cnn_forward(Cnn* c, i64 n, const u64* states_dev, float* logp, float* logv, float* probs, float* value,
                hipStream_t st);
// evaluate the compacted leaf list of a split search round: reads d.eval_list/eval_count/leaf_state,
// writes d.eval_probs/eval_value for those slots.
int cnn_eval_leaves(Cnn* c, const Dev& d, hipStream_t st);

}  // namespace az
