/*
 * az_oracle_cnn.c -- CPU restatement of the Connect4 policy/value CNN forward pass.
 *
 *   TEST INFRASTRUCTURE ONLY (see az_oracle.c).  Batch-1, fp32, eval mode -- exactly what the
 *   reference runs once per playout (MCTS.py:145-146 -> Connect4/Network.py:78-85).
 *
 * Parity status: PINNED against golden (state -> log_prob, value_log_prob, probs, value) vectors
 * produced by the reference's torch module with params/AZ_Connect4_CNN_current.pt
 * (tests/golden/make_fixtures.py), tolerance 1e-4 relative as BASELINE.json:north_star states.
 *
 * Weight blob = every float32 tensor of the reference state_dict, concatenated in state_dict order
 * (num_batches_tracked skipped): 500106 floats.  BatchNorm is applied literally (not folded) here,
 * so this file also checks the BN folding done by the product's az_net_load().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "az_oracle.h"

#define HD 128
#define NP 42
#define BN_EPS 1e-5f /* torch.nn.BatchNorm2d default */

typedef struct {
    const float *g, *b, *m, *v;
} bn_t;

typedef struct {
    float *blob;
    /* transposed conv weights: [tap][ci][co] */
    float *stem_wt;      /* 9*3*128 */
    float *c3_wt[3];     /* 9*128*128 */
    float *c1_wt[3];     /* 128*128 [ci][co] */
    bn_t stem_bn, bn1[3], bn2[3], pbn, vbn;
    const float *pconv, *plin_w, *plin_b; /* (3,128) (7,126) (7) */
    const float *vconv, *vlin_w, *vlin_b; /* (1,128) (3,42) (3) */
} cnn_t;

static const float *take(const float **p, size_t n) {
    const float *r = *p;
    *p += n;
    return r;
}
static bn_t take_bn(const float **p, size_t c) {
    bn_t b;
    b.g = take(p, c);
    b.b = take(p, c);
    b.m = take(p, c);
    b.v = take(p, c);
    return b;
}

/* OIHW (co, ci, kh, kw) -> [tap][ci][co] */
static float *transpose_w(const float *w, int co, int ci, int taps) {
    float *t = (float *)malloc(sizeof(float) * (size_t)co * ci * taps);
    for (int o = 0; o < co; o++)
        for (int i = 0; i < ci; i++)
            for (int k = 0; k < taps; k++) t[((size_t)k * ci + i) * co + o] = w[((size_t)o * ci + i) * taps + k];
    return t;
}

void *orc_cnn_new(const float *blob, int64_t n_floats) {
    if (n_floats != ORC_CNN_BLOB_FLOATS) return NULL;
    cnn_t *n = (cnn_t *)calloc(1, sizeof(cnn_t));
    n->blob = (float *)malloc(sizeof(float) * (size_t)n_floats);
    memcpy(n->blob, blob, sizeof(float) * (size_t)n_floats);
    const float *p = n->blob;
    const float *w = take(&p, 128 * 3 * 9); /* hidden.0 : Network.py:33 */
    n->stem_wt = transpose_w(w, 128, 3, 9);
    n->stem_bn = take_bn(&p, 128);
    for (int b = 0; b < 3; b++) { /* hidden.4/5/6 = Block : Network.py:16-27 */
        w = take(&p, 128 * 128 * 9);
        n->c3_wt[b] = transpose_w(w, 128, 128, 9);
        n->bn1[b] = take_bn(&p, 128);
        w = take(&p, 128 * 128);
        n->c1_wt[b] = transpose_w(w, 128, 128, 1);
        n->bn2[b] = take_bn(&p, 128);
    }
    n->pconv = take(&p, 3 * 128); /* policy_head : Network.py:40-45 */
    n->pbn = take_bn(&p, 3);
    n->plin_w = take(&p, 7 * 126);
    n->plin_b = take(&p, 7);
    n->vconv = take(&p, 128); /* value_head : Network.py:46-51 */
    n->vbn = take_bn(&p, 1);
    n->vlin_w = take(&p, 3 * 42);
    n->vlin_b = take(&p, 3);
    return n;
}

void orc_cnn_free(void *net) {
    cnn_t *n = (cnn_t *)net;
    if (!n) return;
    free(n->stem_wt);
    for (int b = 0; b < 3; b++) {
        free(n->c3_wt[b]);
        free(n->c1_wt[b]);
    }
    free(n->blob);
    free(n);
}

static inline float silu(float x) { return x / (1.0f + expf(-x)); }
static inline float bn_apply(const bn_t *bn, int c, float x) {
    return (x - bn->m[c]) / sqrtf(bn->v[c] + BN_EPS) * bn->g[c] + bn->b[c];
}

/* 3x3, zero padding 1, input [p][ci] (NHWC), output [p][co]; inner loop over co vectorises. */
static void conv3(const float *in, int ci_n, const float *wt, float *out) {
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 7; c++) {
            float *o = out + (size_t)(r * 7 + c) * HD;
            for (int k = 0; k < HD; k++) o[k] = 0.f;
            for (int dr = -1; dr <= 1; dr++)
                for (int dc = -1; dc <= 1; dc++) {
                    int rr = r + dr, cc = c + dc;
                    if (rr < 0 || rr >= 6 || cc < 0 || cc >= 7) continue;
                    int tap = (dr + 1) * 3 + (dc + 1);
                    const float *x = in + (size_t)(rr * 7 + cc) * ci_n;
                    const float *w = wt + (size_t)tap * ci_n * HD;
                    for (int i = 0; i < ci_n; i++) {
                        float xi = x[i];
                        const float *wi = w + (size_t)i * HD;
                        for (int k = 0; k < HD; k++) o[k] += xi * wi[k];
                    }
                }
        }
}

static void conv1(const float *in, const float *wt, float *out) {
    for (int p = 0; p < NP; p++) {
        float *o = out + (size_t)p * HD;
        const float *x = in + (size_t)p * HD;
        for (int k = 0; k < HD; k++) o[k] = 0.f;
        for (int i = 0; i < HD; i++) {
            float xi = x[i];
            const float *wi = wt + (size_t)i * HD;
            for (int k = 0; k < HD; k++) o[k] += xi * wi[k];
        }
    }
}

static void log_softmax(float *x, int n) {
    float mx = x[0], s = 0.f;
    for (int i = 1; i < n; i++)
        if (x[i] > mx) mx = x[i];
    for (int i = 0; i < n; i++) s += expf(x[i] - mx);
    float l = logf(s);
    for (int i = 0; i < n; i++) x[i] = x[i] - mx - l;
}

/* Network.py:60-64 (forward), eval mode: Dropout2d = identity, BN uses running stats. */
void orc_cnn_forward(const void *net, const float *state, float *logp, float *logv) {
    const cnn_t *n = (const cnn_t *)net;
    float in0[NP * 3], x[NP * HD], t[NP * HD], u[NP * HD];
    for (int p = 0; p < NP; p++)
        for (int c = 0; c < 3; c++) in0[p * 3 + c] = state[c * NP + p];
    conv3(in0, 3, n->stem_wt, x);
    for (int p = 0; p < NP; p++)
        for (int c = 0; c < HD; c++) x[p * HD + c] = silu(bn_apply(&n->stem_bn, c, x[p * HD + c]));
    for (int b = 0; b < 3; b++) {
        conv3(x, HD, n->c3_wt[b], t);
        for (int p = 0; p < NP; p++)
            for (int c = 0; c < HD; c++) t[p * HD + c] = silu(bn_apply(&n->bn1[b], c, t[p * HD + c]));
        conv1(t, n->c1_wt[b], u);
        for (int p = 0; p < NP; p++)
            for (int c = 0; c < HD; c++) x[p * HD + c] = silu(x[p * HD + c] + bn_apply(&n->bn2[b], c, u[p * HD + c]));
    }
    float ph[126], vh[42];
    for (int p = 0; p < NP; p++) {
        for (int c = 0; c < 3; c++) {
            float s = 0.f;
            for (int i = 0; i < HD; i++) s += x[p * HD + i] * n->pconv[c * HD + i];
            ph[c * NP + p] = silu(bn_apply(&n->pbn, c, s)); /* nn.Flatten of NCHW: c*42 + h*7 + w */
        }
        float s = 0.f;
        for (int i = 0; i < HD; i++) s += x[p * HD + i] * n->vconv[i];
        vh[p] = silu(bn_apply(&n->vbn, 0, s));
    }
    for (int a = 0; a < 7; a++) {
        float s = n->plin_b[a];
        for (int i = 0; i < 126; i++) s += ph[i] * n->plin_w[a * 126 + i];
        logp[a] = s;
    }
    for (int k = 0; k < 3; k++) {
        float s = n->vlin_b[k];
        for (int i = 0; i < 42; i++) s += vh[i] * n->vlin_w[k * 42 + i];
        logv[k] = s;
    }
    log_softmax(logp, 7);
    log_softmax(logv, 3);
}

/* Network.py:78-85 (predict): classes [draw, mover wins, mover loses]; colour-dependent contempt. */
void orc_cnn_predict(const void *net, const float *state, float *probs, float *value) {
    float logp[7], logv[3], vp[3];
    orc_cnn_forward(net, state, logp, logv);
    for (int a = 0; a < 7; a++) probs[a] = expf(logp[a]);
    for (int k = 0; k < 3; k++) vp[k] = expf(logv[k]);
    float player = state[2 * NP];
    *value = (-player) * 0.5f * vp[0] + vp[1] - vp[2];
}
