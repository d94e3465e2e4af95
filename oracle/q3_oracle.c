/*
 * q3_oracle.c — CPU ORACLE (fp32) for the Qwen3-TTS hot path.  TEST INFRASTRUCTURE ONLY:
 * see q3_oracle.h for who may use it and for the "parity unpinned" statement.
 *
 * Reference citations are into /root/reference (leaxer-ai/leaxer-qwen3-tts v0.2.0).
 * [HINT] marks architecture facts that are NOT in the reference (its networks are opaque
 * .onnx files); they follow the public Qwen3 / Qwen3-Omni model definitions and are pinned
 * against the installed `transformers` code by tests/golden/make_hf_goldens.py.
 */
#define _GNU_SOURCE
#include "q3_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- host-logic constants: src/tts_onnx.h:40-62 ---- */
#define TTS_BOS 151672
#define TTS_EOS 151673
#define TTS_PAD 151671
#define CODEC_BOS 2149
#define CODEC_PAD 2148
#define CODEC_THINK 2154
#define CODEC_NOTHINK 2155
#define CODEC_THINK_BOS 2156
#define CODEC_THINK_EOS 2157
#define LANG_ENGLISH 2050

static char g_err[512];
const char* q3o_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

void q3o_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* primitives                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* fp32 dot product, 16 interleaved partial sums (fixed order, vectorisable without -ffast-math) */
static float dotf(const float* a, const float* b, int n) {
    float acc[16] = {0};
    int k = 0;
    for (; k + 16 <= n; k += 16)
        for (int j = 0; j < 16; ++j) acc[j] += a[k + j] * b[k + j];
    for (int j = 0; k < n; ++k, ++j) acc[j] += a[k] * b[k];
    for (int s = 8; s >= 1; s >>= 1)
        for (int j = 0; j < s; ++j) acc[j] += acc[j + s];
    return acc[0];
}

/* Y[m][n] = bias[n] + sum_k X[m][k] W[n][k]   (W row-major [N][K], i.e. nn.Linear.weight) */
static void linear(const float* X, int M, int K, const float* W, const float* bias, int N, float* Y, int ldy) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const float* w = W + (size_t)n * K;
        for (int m = 0; m < M; ++m) {
            float v = dotf(X + (size_t)m * K, w, K);
            Y[(size_t)m * ldy + n] = bias ? v + bias[n] : v;
        }
    }
}

/* RMSNorm: w * (x * rsqrt(mean(x^2) + eps))   [HINT transformers qwen3 RMSNorm] */
static void rmsnorm(const float* x, const float* w, int n, float eps, float* y) {
    float ss = 0.f;
    for (int i = 0; i < n; ++i) ss += x[i] * x[i];
    float r = 1.0f / sqrtf(ss / (float)n + eps);
    for (int i = 0; i < n; ++i) y[i] = w[i] * (x[i] * r);
}

static inline float siluf(float x) { return x / (1.0f + expf(-x)); }
static inline float geluf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

/* ------------------------------------------------------------------------------------------ */
/* model containers                                                                           */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    float *in_norm, *q, *k, *v, *o, *q_norm, *k_norm, *post_norm, *gate, *up, *down;
    float *attn_scale, *mlp_scale; /* codec pre-transformer LayerScale only */
} layer_w;

typedef struct {
    int H, L, nq, nkv, d, ffn;
    float theta, eps;
    int window;   /* 0 = full causal */
    int qk_norm;  /* per-head q/k RMSNorm (talker, predictor) */
    int lscale;   /* LayerScale on both residual branches (codec pre-transformer) */
    int kv_bf16;  /* K / V rows rounded to bf16 (RNE) as they enter the cache; attention runs in fp32 on the rounded rows (talker only) */
} dec_dims;

typedef struct { float *alpha, *beta; } snake_w;
typedef struct { float *w, *b; int cin, cout, k; float* wt; /* [tap][cout][cin] */ } conv_w;
typedef struct { snake_w a1, a2; conv_w c1, c2; } resunit_w;
typedef struct { snake_w act; conv_w tconv; resunit_w res[3]; } block_w;
typedef struct { conv_w tconv; conv_w dw; float *ln_w, *ln_b, *pw1_w, *pw1_b, *pw2_w, *pw2_b, *gamma; } upstage_w;

struct q3o_model {
    q3o_config c;
    int max_ctx;
    /* talker */
    layer_w* tl;
    float *t_norm, *codec_head, *codec_embed;
    float *text_embed, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    float *kc, *vc; /* [L][nkv][max_ctx][d] */
    int pos;        /* tokens in the talker KV cache */
    int kv_bf16;    /* q3o_set_kv_bf16: talker K / V rows rounded to bf16 on append */
    /* predictor */
    layer_w* pl;
    float* p_norm;
    float **p_head, **p_embed; /* [G-1] x [SV][Hc] heads, [SV][H] embeddings (talker width) */
    float *p_proj_w, *p_proj_b; /* talker width -> predictor width (cp_hidden != hidden: the 1.7B export), else NULL */
    float *pkc, *pvc;          /* [cp_layers][cp_kv][32][d] */
    /* codec decoder */
    layer_w* cl;
    float *c_norm, *code_embed;
    upstage_w up[4];
    conv_w conv_in, conv_out;
    block_w blk[8];
    snake_w snake_out;
    /* speaker encoder (ECAPA-TDNN), torch Conv1d layout [out][in][k] */
    conv_w s_tdnn0, s_tdnn1[3], s_res[3][16], s_tdnn2[3], s_se1[3], s_se2[3], s_mfa, s_asp_tdnn, s_asp_conv, s_fc;
    /* host-logic state (tts_onnx.h:182-186) */
    float *last_hidden, *trailing, *tts_pad;
    int trailing_len;
};

/* predictor width: the 0.6B predictor runs at the talker's width; the 1.7B one is narrower and puts a Linear (with bias) in front of
 * its layers [HINT: Qwen3-TTS `small_to_mtp_projection`, applied to every input row of the predictor; the reference's
 * code_predictor.onnx would hold it inside the graph, its I/O contract (tts_onnx.cpp:734-757) is unchanged] */
static int cp_width(const q3o_config* c) { return c->cp_hidden > 0 ? c->cp_hidden : c->hidden; }

static float* zalloc(size_t n) {
    float* p = (float*)calloc(n ? n : 1, sizeof(float));
    if (!p) { fprintf(stderr, "q3_oracle: out of memory (%zu floats)\n", n); abort(); }
    return p;
}

static void alloc_layers(layer_w* L, int n, int H, int nq, int nkv, int d, int ffn, int qk, int ls) {
    for (int i = 0; i < n; ++i) {
        L[i].in_norm = zalloc(H); L[i].post_norm = zalloc(H);
        L[i].q = zalloc((size_t)nq * d * H); L[i].k = zalloc((size_t)nkv * d * H); L[i].v = zalloc((size_t)nkv * d * H);
        L[i].o = zalloc((size_t)H * nq * d);
        L[i].q_norm = qk ? zalloc(d) : NULL; L[i].k_norm = qk ? zalloc(d) : NULL;
        L[i].gate = zalloc((size_t)ffn * H); L[i].up = zalloc((size_t)ffn * H); L[i].down = zalloc((size_t)H * ffn);
        L[i].attn_scale = ls ? zalloc(H) : NULL; L[i].mlp_scale = ls ? zalloc(H) : NULL;
    }
}
static void free_layers(layer_w* L, int n) {
    if (!L) return;
    for (int i = 0; i < n; ++i) {
        free(L[i].in_norm); free(L[i].post_norm); free(L[i].q); free(L[i].k); free(L[i].v); free(L[i].o);
        free(L[i].q_norm); free(L[i].k_norm); free(L[i].gate); free(L[i].up); free(L[i].down);
        free(L[i].attn_scale); free(L[i].mlp_scale);
    }
    free(L);
}
static void alloc_conv(conv_w* c, int cin, int cout, int k, int depthwise) {
    c->cin = cin; c->cout = cout; c->k = k;
    c->w = zalloc(depthwise ? (size_t)cout * k : (size_t)cin * cout * k);
    c->b = zalloc(cout);
    c->wt = depthwise ? NULL : zalloc((size_t)cin * cout * k);
}
static void free_conv(conv_w* c) { free(c->w); free(c->b); free(c->wt); }
static void alloc_snake(snake_w* s, int c) { s->alpha = zalloc(c); s->beta = zalloc(c); }

q3o_model* q3o_create(const q3o_config* cfg, int max_ctx) {
    q3o_model* m = (q3o_model*)calloc(1, sizeof *m);
    m->c = *cfg; m->max_ctx = max_ctx;
    const q3o_config* c = &m->c;
    int H = c->hidden, G = c->n_groups;
    m->tl = (layer_w*)calloc(c->n_layers, sizeof(layer_w));
    alloc_layers(m->tl, c->n_layers, H, c->n_heads, c->n_kv_heads, c->head_dim, c->ffn, 1, 0);
    m->t_norm = zalloc(H); m->codec_head = zalloc((size_t)c->vocab * H); m->codec_embed = zalloc((size_t)c->vocab * H);
    m->text_embed = zalloc((size_t)c->text_vocab * c->text_hidden);
    m->fc1_w = zalloc((size_t)c->text_hidden * c->text_hidden); m->fc1_b = zalloc(c->text_hidden);
    m->fc2_w = zalloc((size_t)H * c->text_hidden); m->fc2_b = zalloc(H);
    size_t kvn = (size_t)c->n_layers * c->n_kv_heads * max_ctx * c->head_dim;
    m->kc = zalloc(kvn); m->vc = zalloc(kvn);
    m->pl = (layer_w*)calloc(c->cp_layers, sizeof(layer_w));
    int Hc = cp_width(c);
    alloc_layers(m->pl, c->cp_layers, Hc, c->cp_heads, c->cp_kv_heads, c->cp_head_dim, c->cp_ffn, 1, 0);
    m->p_norm = zalloc(Hc);
    if (Hc != H) { m->p_proj_w = zalloc((size_t)Hc * H); m->p_proj_b = zalloc(Hc); }
    m->p_head = (float**)calloc(G, sizeof(float*)); m->p_embed = (float**)calloc(G, sizeof(float*));
    for (int j = 0; j < G - 1; ++j) { m->p_head[j] = zalloc((size_t)c->sub_vocab * Hc); m->p_embed[j] = zalloc((size_t)c->sub_vocab * H); }
    size_t pkvn = (size_t)c->cp_layers * c->cp_kv_heads * 32 * c->cp_head_dim;
    m->pkc = zalloc(pkvn); m->pvc = zalloc(pkvn);
    int CH = c->cd_hidden;
    m->cl = (layer_w*)calloc(c->cd_layers, sizeof(layer_w));
    alloc_layers(m->cl, c->cd_layers, CH, c->cd_heads, c->cd_heads, c->cd_head_dim, c->cd_ffn, 0, 1);
    m->c_norm = zalloc(CH); m->code_embed = zalloc((size_t)G * c->cd_codebook * CH);
    for (int s = 0; s < c->cd_n_up; ++s) {
        upstage_w* u = &m->up[s];
        int f = c->cd_up_ratios[s];
        alloc_conv(&u->tconv, CH, CH, f, 0); alloc_conv(&u->dw, CH, CH, 7, 1);
        u->ln_w = zalloc(CH); u->ln_b = zalloc(CH);
        u->pw1_w = zalloc((size_t)4 * CH * CH); u->pw1_b = zalloc(4 * CH);
        u->pw2_w = zalloc((size_t)4 * CH * CH); u->pw2_b = zalloc(CH); u->gamma = zalloc(CH);
    }
    int D = c->cd_decoder_dim;
    alloc_conv(&m->conv_in, CH, D, 7, 0);
    for (int i = 0; i < c->cd_n_blocks; ++i) {
        int cin = D >> i, cout = D >> (i + 1), r = c->cd_up_rates[i];
        alloc_snake(&m->blk[i].act, cin);
        alloc_conv(&m->blk[i].tconv, cin, cout, 2 * r, 0);
        for (int u = 0; u < 3; ++u) {
            alloc_snake(&m->blk[i].res[u].a1, cout); alloc_snake(&m->blk[i].res[u].a2, cout);
            alloc_conv(&m->blk[i].res[u].c1, cout, cout, 7, 0); alloc_conv(&m->blk[i].res[u].c2, cout, cout, 1, 0);
        }
    }
    int OD = D >> c->cd_n_blocks;
    alloc_snake(&m->snake_out, OD);
    alloc_conv(&m->conv_out, OD, 1, 7, 0);
    if (c->spk_enc_dim > 0) {
        int SC = c->spk_channels, sub = SC / c->spk_scale;
        alloc_conv(&m->s_tdnn0, c->spk_mel, SC, 5, 0);
        for (int i = 0; i < 3; ++i) {
            alloc_conv(&m->s_tdnn1[i], SC, SC, 1, 0);
            for (int j = 0; j < c->spk_scale - 1; ++j) alloc_conv(&m->s_res[i][j], sub, sub, 3, 0);
            alloc_conv(&m->s_tdnn2[i], SC, SC, 1, 0);
            alloc_conv(&m->s_se1[i], SC, c->spk_se, 1, 0);
            alloc_conv(&m->s_se2[i], c->spk_se, SC, 1, 0);
        }
        alloc_conv(&m->s_mfa, 3 * SC, 3 * SC, 1, 0);
        alloc_conv(&m->s_asp_tdnn, 9 * SC, c->spk_att, 1, 0);
        alloc_conv(&m->s_asp_conv, c->spk_att, 3 * SC, 1, 0);
        alloc_conv(&m->s_fc, 6 * SC, c->spk_enc_dim, 1, 0);
    }
    m->last_hidden = zalloc(H); m->tts_pad = zalloc(H); m->trailing = NULL; m->trailing_len = 0;
    return m;
}

void q3o_destroy(q3o_model* m) {
    if (!m) return;
    const q3o_config* c = &m->c;
    free_layers(m->tl, c->n_layers); free_layers(m->pl, c->cp_layers); free_layers(m->cl, c->cd_layers);
    free(m->t_norm); free(m->codec_head); free(m->codec_embed); free(m->text_embed);
    free(m->fc1_w); free(m->fc1_b); free(m->fc2_w); free(m->fc2_b); free(m->kc); free(m->vc);
    free(m->p_norm); free(m->p_proj_w); free(m->p_proj_b);
    for (int j = 0; j < c->n_groups - 1; ++j) { free(m->p_head[j]); free(m->p_embed[j]); }
    free(m->p_head); free(m->p_embed); free(m->pkc); free(m->pvc);
    free(m->c_norm); free(m->code_embed);
    for (int s = 0; s < c->cd_n_up; ++s) {
        upstage_w* u = &m->up[s];
        free_conv(&u->tconv); free_conv(&u->dw); free(u->ln_w); free(u->ln_b);
        free(u->pw1_w); free(u->pw1_b); free(u->pw2_w); free(u->pw2_b); free(u->gamma);
    }
    free_conv(&m->conv_in); free_conv(&m->conv_out);
    for (int i = 0; i < c->cd_n_blocks; ++i) {
        free(m->blk[i].act.alpha); free(m->blk[i].act.beta); free_conv(&m->blk[i].tconv);
        for (int u = 0; u < 3; ++u) {
            resunit_w* r = &m->blk[i].res[u];
            free(r->a1.alpha); free(r->a1.beta); free(r->a2.alpha); free(r->a2.beta);
            free_conv(&r->c1); free_conv(&r->c2);
        }
    }
    free(m->snake_out.alpha); free(m->snake_out.beta);
    if (c->spk_enc_dim > 0) {
        free_conv(&m->s_tdnn0); free_conv(&m->s_mfa); free_conv(&m->s_asp_tdnn); free_conv(&m->s_asp_conv); free_conv(&m->s_fc);
        for (int i = 0; i < 3; ++i) {
            free_conv(&m->s_tdnn1[i]); free_conv(&m->s_tdnn2[i]); free_conv(&m->s_se1[i]); free_conv(&m->s_se2[i]);
            for (int j = 0; j < c->spk_scale - 1; ++j) free_conv(&m->s_res[i][j]);
        }
    }
    free(m->last_hidden); free(m->tts_pad); free(m->trailing);
    free(m);
}

/* ------------------------------------------------------------------------------------------ */
/* tensor registry: name -> (pointer, numel).  Names are shared with the product's blob.     */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float* p; int64_t n; conv_w* conv; int tconv; } slot;

static int layer_slot(layer_w* L, const char* f, int H, int nq, int nkv, int d, int ffn, slot* s) {
    if (!strcmp(f, "input_norm")) { s->p = L->in_norm; s->n = H; }
    else if (!strcmp(f, "post_norm")) { s->p = L->post_norm; s->n = H; }
    else if (!strcmp(f, "q_proj")) { s->p = L->q; s->n = (int64_t)nq * d * H; }
    else if (!strcmp(f, "k_proj")) { s->p = L->k; s->n = (int64_t)nkv * d * H; }
    else if (!strcmp(f, "v_proj")) { s->p = L->v; s->n = (int64_t)nkv * d * H; }
    else if (!strcmp(f, "o_proj")) { s->p = L->o; s->n = (int64_t)H * nq * d; }
    else if (!strcmp(f, "q_norm")) { s->p = L->q_norm; s->n = d; }
    else if (!strcmp(f, "k_norm")) { s->p = L->k_norm; s->n = d; }
    else if (!strcmp(f, "gate_proj")) { s->p = L->gate; s->n = (int64_t)ffn * H; }
    else if (!strcmp(f, "up_proj")) { s->p = L->up; s->n = (int64_t)ffn * H; }
    else if (!strcmp(f, "down_proj")) { s->p = L->down; s->n = (int64_t)H * ffn; }
    else if (!strcmp(f, "attn_scale")) { s->p = L->attn_scale; s->n = H; }
    else if (!strcmp(f, "mlp_scale")) { s->p = L->mlp_scale; s->n = H; }
    else return -1;
    return s->p ? 0 : -1;
}
static int conv_slot(conv_w* c, const char* f, int depthwise, int tconv, slot* s) {
    if (!strcmp(f, "w")) { s->p = c->w; s->n = depthwise ? (int64_t)c->cout * c->k : (int64_t)c->cin * c->cout * c->k; s->conv = depthwise ? NULL : c; s->tconv = tconv; return 0; }
    if (!strcmp(f, "b")) { s->p = c->b; s->n = c->cout; return 0; }
    return -1;
}
static int snake_slot(snake_w* a, const char* f, int ch, slot* s) {
    if (!strcmp(f, "alpha")) { s->p = a->alpha; s->n = ch; return 0; }
    if (!strcmp(f, "beta")) { s->p = a->beta; s->n = ch; return 0; }
    return -1;
}

/* name == prefix + decimal index */
static int idx_suffix(const char* name, const char* prefix, int* idx) {
    size_t n = strlen(prefix);
    if (strncmp(name, prefix, n) || !name[n]) return 0;
    int v = 0;
    for (const char* p = name + n; *p; ++p) { if (*p < '0' || *p > '9') return 0; v = v * 10 + (*p - '0'); }
    *idx = v;
    return 1;
}

static int resolve(q3o_model* m, const char* name, slot* s) {
    const q3o_config* c = &m->c;
    int H = c->hidden, i, j, u; char f[64];
    memset(s, 0, sizeof *s);
    if (sscanf(name, "talker.layers.%d.%63s", &i, f) == 2 && i >= 0 && i < c->n_layers)
        return layer_slot(&m->tl[i], f, H, c->n_heads, c->n_kv_heads, c->head_dim, c->ffn, s);
    if (!strcmp(name, "talker.norm")) { s->p = m->t_norm; s->n = H; return 0; }
    if (!strcmp(name, "talker.codec_head")) { s->p = m->codec_head; s->n = (int64_t)c->vocab * H; return 0; }
    if (!strcmp(name, "talker.codec_embed")) { s->p = m->codec_embed; s->n = (int64_t)c->vocab * H; return 0; }
    if (!strcmp(name, "text.embed")) { s->p = m->text_embed; s->n = (int64_t)c->text_vocab * c->text_hidden; return 0; }
    if (!strcmp(name, "text.fc1.w")) { s->p = m->fc1_w; s->n = (int64_t)c->text_hidden * c->text_hidden; return 0; }
    if (!strcmp(name, "text.fc1.b")) { s->p = m->fc1_b; s->n = c->text_hidden; return 0; }
    if (!strcmp(name, "text.fc2.w")) { s->p = m->fc2_w; s->n = (int64_t)H * c->text_hidden; return 0; }
    if (!strcmp(name, "text.fc2.b")) { s->p = m->fc2_b; s->n = H; return 0; }
    if (sscanf(name, "cp.layers.%d.%63s", &i, f) == 2 && i >= 0 && i < c->cp_layers)
        return layer_slot(&m->pl[i], f, cp_width(c), c->cp_heads, c->cp_kv_heads, c->cp_head_dim, c->cp_ffn, s);
    if (!strcmp(name, "cp.norm")) { s->p = m->p_norm; s->n = cp_width(c); return 0; }
    if (!strcmp(name, "cp.proj.w") && m->p_proj_w) { s->p = m->p_proj_w; s->n = (int64_t)cp_width(c) * H; return 0; }
    if (!strcmp(name, "cp.proj.b") && m->p_proj_b) { s->p = m->p_proj_b; s->n = cp_width(c); return 0; }
    if (idx_suffix(name, "cp.head.", &j) && j < c->n_groups - 1) { s->p = m->p_head[j]; s->n = (int64_t)c->sub_vocab * cp_width(c); return 0; }
    if (idx_suffix(name, "cp.embed.", &j) && j < c->n_groups - 1) { s->p = m->p_embed[j]; s->n = (int64_t)c->sub_vocab * H; return 0; }
    int CH = c->cd_hidden;
    if (sscanf(name, "cd.layers.%d.%63s", &i, f) == 2 && i >= 0 && i < c->cd_layers)
        return layer_slot(&m->cl[i], f, CH, c->cd_heads, c->cd_heads, c->cd_head_dim, c->cd_ffn, s);
    if (!strcmp(name, "cd.norm")) { s->p = m->c_norm; s->n = CH; return 0; }
    if (!strcmp(name, "cd.code_embed")) { s->p = m->code_embed; s->n = (int64_t)c->n_groups * c->cd_codebook * CH; return 0; }
    if (sscanf(name, "cd.up.%d.tconv.%63s", &i, f) == 2 && i >= 0 && i < c->cd_n_up) return conv_slot(&m->up[i].tconv, f, 0, 1, s);
    if (sscanf(name, "cd.up.%d.cnx.dw.%63s", &i, f) == 2 && i >= 0 && i < c->cd_n_up) return conv_slot(&m->up[i].dw, f, 1, 0, s);
    if (sscanf(name, "cd.up.%d.cnx.%63s", &i, f) == 2 && i >= 0 && i < c->cd_n_up) {
        upstage_w* us = &m->up[i];
        if (!strcmp(f, "ln.w")) { s->p = us->ln_w; s->n = CH; return 0; }
        if (!strcmp(f, "ln.b")) { s->p = us->ln_b; s->n = CH; return 0; }
        if (!strcmp(f, "pw1.w")) { s->p = us->pw1_w; s->n = (int64_t)4 * CH * CH; return 0; }
        if (!strcmp(f, "pw1.b")) { s->p = us->pw1_b; s->n = 4 * CH; return 0; }
        if (!strcmp(f, "pw2.w")) { s->p = us->pw2_w; s->n = (int64_t)4 * CH * CH; return 0; }
        if (!strcmp(f, "pw2.b")) { s->p = us->pw2_b; s->n = CH; return 0; }
        if (!strcmp(f, "gamma")) { s->p = us->gamma; s->n = CH; return 0; }
        return -1;
    }
    if (c->spk_enc_dim > 0 && !strncmp(name, "spk.", 4)) {
        if (sscanf(name, "spk.tdnn0.%63s", f) == 1) return conv_slot(&m->s_tdnn0, f, 0, 0, s);
        if (sscanf(name, "spk.blocks.%d.res2net.%d.%63s", &i, &j, f) == 3 && i >= 0 && i < 3 && j >= 0 && j < c->spk_scale - 1)
            return conv_slot(&m->s_res[i][j], f, 0, 0, s);
        if (sscanf(name, "spk.blocks.%d.tdnn1.%63s", &i, f) == 2 && i >= 0 && i < 3) return conv_slot(&m->s_tdnn1[i], f, 0, 0, s);
        if (sscanf(name, "spk.blocks.%d.tdnn2.%63s", &i, f) == 2 && i >= 0 && i < 3) return conv_slot(&m->s_tdnn2[i], f, 0, 0, s);
        if (sscanf(name, "spk.blocks.%d.se1.%63s", &i, f) == 2 && i >= 0 && i < 3) return conv_slot(&m->s_se1[i], f, 0, 0, s);
        if (sscanf(name, "spk.blocks.%d.se2.%63s", &i, f) == 2 && i >= 0 && i < 3) return conv_slot(&m->s_se2[i], f, 0, 0, s);
        if (sscanf(name, "spk.mfa.%63s", f) == 1) return conv_slot(&m->s_mfa, f, 0, 0, s);
        if (sscanf(name, "spk.asp.tdnn.%63s", f) == 1) return conv_slot(&m->s_asp_tdnn, f, 0, 0, s);
        if (sscanf(name, "spk.asp.conv.%63s", f) == 1) return conv_slot(&m->s_asp_conv, f, 0, 0, s);
        if (sscanf(name, "spk.fc.%63s", f) == 1) return conv_slot(&m->s_fc, f, 0, 0, s);
        return -1;
    }
    if (sscanf(name, "cd.dec.conv_in.%63s", f) == 1) return conv_slot(&m->conv_in, f, 0, 0, s);
    if (sscanf(name, "cd.dec.conv_out.%63s", f) == 1) return conv_slot(&m->conv_out, f, 0, 0, s);
    if (sscanf(name, "cd.dec.snake_out.%63s", f) == 1) return snake_slot(&m->snake_out, f, c->cd_decoder_dim >> c->cd_n_blocks, s);
    if (sscanf(name, "cd.dec.blocks.%d.res.%d.%63s", &i, &u, f) == 3 && i >= 0 && i < c->cd_n_blocks && u >= 0 && u < 3) {
        resunit_w* r = &m->blk[i].res[u];
        int ch = c->cd_decoder_dim >> (i + 1);
        if (!strncmp(f, "act1.", 5)) return snake_slot(&r->a1, f + 5, ch, s);
        if (!strncmp(f, "act2.", 5)) return snake_slot(&r->a2, f + 5, ch, s);
        if (!strncmp(f, "conv1.", 6)) return conv_slot(&r->c1, f + 6, 0, 0, s);
        if (!strncmp(f, "conv2.", 6)) return conv_slot(&r->c2, f + 6, 0, 0, s);
        return -1;
    }
    if (sscanf(name, "cd.dec.blocks.%d.%63s", &i, f) == 2 && i >= 0 && i < c->cd_n_blocks) {
        if (!strncmp(f, "snake.", 6)) return snake_slot(&m->blk[i].act, f + 6, c->cd_decoder_dim >> i, s);
        if (!strncmp(f, "tconv.", 6)) return conv_slot(&m->blk[i].tconv, f + 6, 0, 1, s);
        return -1;
    }
    return -1;
}

int64_t q3o_tensor_numel(q3o_model* m, const char* name) {
    slot s;
    if (resolve(m, name, &s) != 0) return -1;
    return s.n;
}

int q3o_set_tensor(q3o_model* m, const char* name, const float* data, int64_t n) {
    slot s;
    if (resolve(m, name, &s) != 0) FAIL("unknown tensor '%s'", name);
    if (s.n != n) FAIL("tensor '%s': expected %lld elements, got %lld", name, (long long)s.n, (long long)n);
    memcpy(s.p, data, (size_t)n * sizeof(float));
    if (s.conv) { /* re-pack to [tap][cout][cin] */
        conv_w* c = s.conv;
        for (int t = 0; t < c->k; ++t)
            for (int co = 0; co < c->cout; ++co)
                for (int ci = 0; ci < c->cin; ++ci)
                    c->wt[((size_t)t * c->cout + co) * c->cin + ci] =
                        s.tconv ? c->w[((size_t)ci * c->cout + co) * c->k + t]   /* ConvTranspose1d: [in][out][k] */
                                : c->w[((size_t)co * c->cin + ci) * c->k + t];   /* Conv1d: [out][in][k] */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* decoder stack  [HINT: transformers modeling_qwen3.py:140-280; qwen3_omni_moe :2250-2380,     */
/* :3267-3450].  x: [M][H] rows at absolute positions pos0..pos0+M-1; K/V appended to kc/vc.   */
/* ------------------------------------------------------------------------------------------ */

/* bf16 KV-cache mode of the product (Q3TTS_FLAG_KV_BF16): the talker's K / V rows are rounded to bf16 (RNE) where they enter the cache,
 * replacing the reference's fp32 std::vector KVCache (src/tts_onnx.h:108-115) by half the bytes; fp32 attention math on the rounded rows. */
void q3o_set_kv_bf16(q3o_model* m, int on) { if (m) m->kv_bf16 = on ? 1 : 0; }

static float bf16_round(float f) {   /* round-to-nearest-even to 8 significant bits (finite inputs), result still an fp32 */
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&f, &u, 4);
    return f;
}

static void rope_row(float* v, int d, int pos, float theta) {
    int half = d / 2;
    for (int i = 0; i < half; ++i) {
        float inv = 1.0f / powf(theta, (float)(2 * i) / (float)d);
        float ang = (float)pos * inv;
        float cs = cosf(ang), sn = sinf(ang);
        float a = v[i], b = v[i + half];
        /* q*cos + rotate_half(q)*sin, rotate_half = [-x2, x1] */
        v[i] = a * cs + (-b) * sn;
        v[i + half] = b * cs + a * sn;
    }
}

static void dec_forward(const dec_dims* D, const layer_w* Ls, float* kc, float* vc, int Tmax,
                        float* x, int M, int pos0) {
    int H = D->H, nq = D->nq, nkv = D->nkv, d = D->d, ffn = D->ffn, grp = nq / nkv;
    float scaling = 1.0f / sqrtf((float)d);
    float* h = zalloc((size_t)M * H);
    float* q = zalloc((size_t)M * nq * d);
    float* k = zalloc((size_t)M * nkv * d);
    float* v = zalloc((size_t)M * nkv * d);
    float* att = zalloc((size_t)M * nq * d);
    float* o = zalloc((size_t)M * H);
    float* g = zalloc((size_t)M * ffn);
    float* u = zalloc((size_t)M * ffn);
    for (int l = 0; l < D->L; ++l) {
        const layer_w* W = &Ls[l];
        for (int m = 0; m < M; ++m) rmsnorm(x + (size_t)m * H, W->in_norm, H, D->eps, h + (size_t)m * H);
        linear(h, M, H, W->q, NULL, nq * d, q, nq * d);
        linear(h, M, H, W->k, NULL, nkv * d, k, nkv * d);
        linear(h, M, H, W->v, NULL, nkv * d, v, nkv * d);
        for (int m = 0; m < M; ++m) {
            int p = pos0 + m;
            for (int hh = 0; hh < nq; ++hh) {
                float* qh = q + ((size_t)m * nq + hh) * d;
                if (D->qk_norm) rmsnorm(qh, W->q_norm, d, D->eps, qh);
                rope_row(qh, d, p, D->theta);
            }
            for (int hh = 0; hh < nkv; ++hh) {
                float* kh = k + ((size_t)m * nkv + hh) * d;
                if (D->qk_norm) rmsnorm(kh, W->k_norm, d, D->eps, kh);
                rope_row(kh, d, p, D->theta);
                float* kdst = kc + (((size_t)l * nkv + hh) * Tmax + p) * d;
                float* vdst = vc + (((size_t)l * nkv + hh) * Tmax + p) * d;
                memcpy(kdst, kh, d * sizeof(float));
                memcpy(vdst, v + ((size_t)m * nkv + hh) * d, d * sizeof(float));
                /* bf16 KV mode (the product's q3tts KV_BF16 flag rounds at the same point): every reader, this token's own step
                 * included, sees the rounded rows */
                if (D->kv_bf16) for (int e = 0; e < d; ++e) { kdst[e] = bf16_round(kdst[e]); vdst[e] = bf16_round(vdst[e]); }
            }
        }
#pragma omp parallel for collapse(2) schedule(static)
        for (int m = 0; m < M; ++m)
            for (int hh = 0; hh < nq; ++hh) {
                int p = pos0 + m, kvh = hh / grp;
                int t0 = (D->window > 0 && p - D->window + 1 > 0) ? p - D->window + 1 : 0; /* keys j: j<=p, p-j<window */
                int n = p - t0 + 1;
                float* sc = (float*)malloc((size_t)n * sizeof(float));
                const float* qh = q + ((size_t)m * nq + hh) * d;
                const float* kb = kc + (((size_t)l * nkv + kvh) * Tmax) * d;
                const float* vb = vc + (((size_t)l * nkv + kvh) * Tmax) * d;
                float mx = -INFINITY;
                for (int t = 0; t < n; ++t) { sc[t] = dotf(qh, kb + (size_t)(t0 + t) * d, d) * scaling; if (sc[t] > mx) mx = sc[t]; }
                float sum = 0.f;
                for (int t = 0; t < n; ++t) { sc[t] = expf(sc[t] - mx); sum += sc[t]; }
                float* oh = att + ((size_t)m * nq + hh) * d;
                for (int e = 0; e < d; ++e) oh[e] = 0.f;
                for (int t = 0; t < n; ++t) {
                    float pw = sc[t] / sum;
                    const float* vr = vb + (size_t)(t0 + t) * d;
                    for (int e = 0; e < d; ++e) oh[e] += pw * vr[e];
                }
                free(sc);
            }
        linear(att, M, nq * d, W->o, NULL, H, o, H);
        for (size_t i = 0; i < (size_t)M * H; ++i) x[i] += D->lscale ? W->attn_scale[i % H] * o[i] : o[i];
        for (int m = 0; m < M; ++m) rmsnorm(x + (size_t)m * H, W->post_norm, H, D->eps, h + (size_t)m * H);
        linear(h, M, H, W->gate, NULL, ffn, g, ffn);
        linear(h, M, H, W->up, NULL, ffn, u, ffn);
        for (size_t i = 0; i < (size_t)M * ffn; ++i) g[i] = siluf(g[i]) * u[i];
        linear(g, M, ffn, W->down, NULL, H, o, H);
        for (size_t i = 0; i < (size_t)M * H; ++i) x[i] += D->lscale ? W->mlp_scale[i % H] * o[i] : o[i];
    }
    free(h); free(q); free(k); free(v); free(att); free(o); free(g); free(u);
}

static dec_dims talker_dims(const q3o_config* c) {
    dec_dims D = { c->hidden, c->n_layers, c->n_heads, c->n_kv_heads, c->head_dim, c->ffn, c->rope_theta, c->rms_eps, 0, 1, 0, 0 };
    return D;
}
static dec_dims cp_dims(const q3o_config* c) {
    dec_dims D = { cp_width(c), c->cp_layers, c->cp_heads, c->cp_kv_heads, c->cp_head_dim, c->cp_ffn, c->cp_rope_theta, c->cp_rms_eps, 0, 1, 0, 0 };
    return D;
}
static dec_dims cd_dims(const q3o_config* c) {
    dec_dims D = { c->cd_hidden, c->cd_layers, c->cd_heads, c->cd_heads, c->cd_head_dim, c->cd_ffn, c->cd_rope_theta, c->cd_rms_eps, c->cd_window, 0, 1, 0 };
    return D;
}

/* ------------------------------------------------------------------------------------------ */
/* session-shaped entry points                                                                */
/* ------------------------------------------------------------------------------------------ */

/* text_project.onnx {input_ids -> embeds} (tts_onnx.cpp:545-559).  [HINT] embedding(text_hidden)
 * -> Linear+bias -> SiLU -> Linear+bias (Qwen3OmniMoeTalkerResizeMLP, modeling :2207-2215). */
int q3o_text_project(q3o_model* m, const int64_t* ids, int n, float* out) {
    const q3o_config* c = &m->c;
    int TH = c->text_hidden, H = c->hidden;
    float* t = zalloc(TH);
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= c->text_vocab) { free(t); FAIL("text id %lld out of range", (long long)ids[i]); }
        linear(m->text_embed + (size_t)ids[i] * TH, 1, TH, m->fc1_w, m->fc1_b, TH, t, TH);
        for (int j = 0; j < TH; ++j) t[j] = siluf(t[j]);
        linear(t, 1, TH, m->fc2_w, m->fc2_b, H, out + (size_t)i * H, H);
    }
    free(t);
    return 0;
}

/* codec_embed.onnx (tts_onnx.cpp:561-590) */
int q3o_codec_embed(q3o_model* m, const int64_t* ids, int n, float* out) {
    int H = m->c.hidden;
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= m->c.vocab) FAIL("codec id %lld out of range", (long long)ids[i]);
        memcpy(out + (size_t)i * H, m->codec_embed + (size_t)ids[i] * H, H * sizeof(float));
    }
    return 0;
}

/* code_predictor_embed.onnx {input_ids, generation_step -> embeds} (tts_onnx.cpp:592-613) */
int q3o_cp_embed(q3o_model* m, int64_t id, int step, float* out) {
    int H = m->c.hidden;
    if (step < 0 || step >= m->c.n_groups - 1 || id < 0 || id >= m->c.sub_vocab) FAIL("cp_embed(%lld,%d) out of range", (long long)id, step);
    memcpy(out, m->p_embed[step] + (size_t)id * H, H * sizeof(float));
    return 0;
}

/* talker_prefill.onnx (tts_onnx.cpp:615-665): logits for all S rows; last_hidden = final-norm
 * hidden of the last row (the reference keeps the first H floats of a last-position-only output,
 * :651-652); K/V kept internally instead of being shuttled through host vectors. */
int q3o_prefill(q3o_model* m, const float* embeds, int S, float* logits, float* last_hidden) {
    const q3o_config* c = &m->c;
    int H = c->hidden;
    if (S <= 0 || S > m->max_ctx) FAIL("prefill length %d out of range", S);
    dec_dims D = talker_dims(c);
    D.kv_bf16 = m->kv_bf16;
    float* x = zalloc((size_t)S * H);
    memcpy(x, embeds, (size_t)S * H * sizeof(float));
    dec_forward(&D, m->tl, m->kc, m->vc, m->max_ctx, x, S, 0);
    m->pos = S;
    float* hn = zalloc((size_t)S * H);
    for (int i = 0; i < S; ++i) rmsnorm(x + (size_t)i * H, m->t_norm, H, c->rms_eps, hn + (size_t)i * H);
    linear(hn, S, H, m->codec_head, NULL, c->vocab, logits, c->vocab);
    memcpy(m->last_hidden, hn + (size_t)(S - 1) * H, H * sizeof(float));
    if (last_hidden) memcpy(last_hidden, m->last_hidden, H * sizeof(float));
    free(x); free(hn);
    return 0;
}

/* talker_decode.onnx (tts_onnx.cpp:667-732) */
int q3o_decode(q3o_model* m, const float* embed, float* logits, float* last_hidden) {
    const q3o_config* c = &m->c;
    int H = c->hidden;
    if (m->pos >= m->max_ctx) FAIL("KV cache full (%d)", m->max_ctx);
    dec_dims D = talker_dims(c);
    D.kv_bf16 = m->kv_bf16;
    float* x = zalloc(H);
    memcpy(x, embed, H * sizeof(float));
    dec_forward(&D, m->tl, m->kc, m->vc, m->max_ctx, x, 1, m->pos);
    m->pos += 1;
    rmsnorm(x, m->t_norm, H, c->rms_eps, m->last_hidden);
    linear(m->last_hidden, 1, H, m->codec_head, NULL, c->vocab, logits, c->vocab);
    if (last_hidden) memcpy(last_hidden, m->last_hidden, H * sizeof(float));
    free(x);
    return 0;
}

/* rows [n][H] at talker width -> predictor input [n][Hc] (caller frees) */
static float* cp_input(q3o_model* m, const float* rows, int n) {
    int H = m->c.hidden, Hc = cp_width(&m->c);
    float* x = zalloc((size_t)n * Hc);
    if (m->p_proj_w) linear(rows, n, H, m->p_proj_w, m->p_proj_b, Hc, x, Hc);
    else memcpy(x, rows, (size_t)n * H * sizeof(float));
    return x;
}

/* code_predictor.onnx {inputs_embeds [1,n,H], generation_step -> logits} (tts_onnx.cpp:734-757):
 * full causal re-run over the n rows, final norm, head #step applied to the LAST row
 * (the reference consumes the first SV floats, :755-756). */
int q3o_code_predictor(q3o_model* m, const float* seq, int n, int step, float* logits) {
    const q3o_config* c = &m->c;
    if (n < 1 || n > 32 || step < 0 || step >= c->n_groups - 1) FAIL("code_predictor(n=%d, step=%d) out of range", n, step);
    dec_dims D = cp_dims(c);
    int Hc = cp_width(c);
    float* x = cp_input(m, seq, n);
    dec_forward(&D, m->pl, m->pkc, m->pvc, 32, x, n, 0);
    float* hn = zalloc(Hc);
    rmsnorm(x + (size_t)(n - 1) * Hc, m->p_norm, Hc, c->cp_rms_eps, hn);
    linear(hn, 1, Hc, m->p_head[step], NULL, c->sub_vocab, logits, c->sub_vocab);
    free(x); free(hn);
    return 0;
}

/* KV-cached variant of the same computation: rows [from, n) are new, cache holds rows [0, from). */
static void cp_cached_step(q3o_model* m, const float* rows, int from, int n, int step, float* logits) {
    const q3o_config* c = &m->c;
    int M = n - from;
    dec_dims D = cp_dims(c);
    int Hc = cp_width(c);
    float* x = cp_input(m, rows, M);
    dec_forward(&D, m->pl, m->pkc, m->pvc, 32, x, M, from);
    float* hn = zalloc(Hc);
    rmsnorm(x + (size_t)(M - 1) * Hc, m->p_norm, Hc, c->cp_rms_eps, hn);
    linear(hn, 1, Hc, m->p_head[step], NULL, c->sub_vocab, logits, c->sub_vocab);
    free(x); free(hn);
}

/* ------------------------------------------------------------------------------------------ */
/* sampler — tts_onnx.cpp:878-950                                                             */
/* ------------------------------------------------------------------------------------------ */

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* The reference draws from a function-static mt19937 seeded by random_device (:901-902), so its
 * sampled output is not reproducible; the restatement substitutes a counter-based generator:
 * one uniform in [0,1) per (seed, utterance stream, frame, codebook group). */
float q3o_rng_uniform(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t group) {
    uint64_t k = mix64(seed ^ mix64(((uint64_t)stream << 32) | frame));
    k = mix64(k + group);
    return (float)(k >> 40) * (1.0f / 16777216.0f);
}

/* exp() of the sampler, fully specified so that the CPU oracle and the HIP sampler (csrc/q3_decode_kernels.hip: q3_expf) agree BIT FOR BIT:
 * libm's expf and the device library's differ in the last place on a few per cent of the inputs, and a probability that differs in its
 * last bit can flip a top-p cut or a draw that lands on a boundary.  Only IEEE-exact operations (mul, fma, rint, scaling by powers of
 * two), in a fixed order; -ffp-contract=off on both sides.  Cody-Waite reduction + the degree-5 Cephes polynomial: within 1 ulp of
 * libm's expf (the reference uses std::exp, tts_onnx.cpp:912). */
float q3o_expf(float x) {
    if (!(x > -103.0f)) return 0.0f;                    /* underflows past the smallest subnormal; also -inf */
    /* domain: x <= 43 (every caller passes x - max <= 0).  The scale below is built as 2^(n+64) * 2^-64, whose exponent field holds
     * n <= 63, i.e. x <= 43.6: larger arguments are clamped here (the result is then a finite lower bound, never inf / a sign flip) */
    if (x > 43.0f) x = 43.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r) + 1.0f;
    /* y * 2^n in two exact-or-correctly-rounded steps: 2^(n+64) is normal for n >= -151, the second factor may round into a subnormal */
    union { uint32_t u; float f; } s1, s2;
    const int ni = (int)n;
    s1.u = (uint32_t)(ni + 64 + 127) << 23;
    s2.u = (uint32_t)(-64 + 127) << 23;
    return (y * s1.f) * s2.f;
}

/* Diagnostic switch (tests/test_cpu_host.py): evaluate the sampler's softmax with libm's expf — what the reference's std::exp
 * (tts_onnx.cpp:912) is on the host — instead of the fully specified q3o_expf the HIP sampler shares.  Off by default; process-global. */
static int g_sampler_exp_libm = 0;
void q3o_set_sampler_exp_libm(int on) { g_sampler_exp_libm = on ? 1 : 0; }

/* :907-915 */
void q3o_softmax(float* x, int n) {
    float mx = x[0];
    for (int i = 1; i < n; ++i) if (x[i] > mx) mx = x[i];
    float sum = 0.f;
    if (g_sampler_exp_libm) for (int i = 0; i < n; ++i) { x[i] = expf(x[i] - mx); sum += x[i]; }
    else for (int i = 0; i < n; ++i) { x[i] = q3o_expf(x[i] - mx); sum += x[i]; }
    for (int i = 0; i < n; ++i) x[i] /= sum;
}
static void softmax_libm(float* x, int n) {
    float mx = x[0];
    for (int i = 1; i < n; ++i) if (x[i] > mx) mx = x[i];
    float sum = 0.f;
    for (int i = 0; i < n; ++i) { x[i] = expf(x[i] - mx); sum += x[i]; }
    for (int i = 0; i < n; ++i) x[i] /= sum;
}

static int cmp_desc(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}
/* :917-927 — threshold = k-th largest value, everything strictly below -> -inf (ties kept) */
void q3o_top_k_filter(float* x, int n, int k) {
    if (k <= 0 || k >= n) return;
    float* s = (float*)malloc((size_t)n * sizeof(float));
    memcpy(s, x, (size_t)n * sizeof(float));
    qsort(s, n, sizeof(float), cmp_desc);
    float thr = s[k - 1];
    free(s);
    for (int i = 0; i < n; ++i) if (x[i] < thr) x[i] = -INFINITY;
}

typedef struct { float p; int i; } pi_t;
static int cmp_pi(const void* a, const void* b) {
    const pi_t *x = (const pi_t*)a, *y = (const pi_t*)b;
    if (x->p != y->p) return (x->p < y->p) - (x->p > y->p);
    return (x->i > y->i) - (x->i < y->i); /* std::sort leaves equal elements unordered; we fix index-ascending */
}
/* :929-950 — sort descending, keep through the first element whose running sum exceeds p */
void q3o_top_p_filter(float* probs, int n, float p) {
    if (p >= 1.0f) return;
    pi_t* idx = (pi_t*)malloc((size_t)n * sizeof(pi_t));
    for (int i = 0; i < n; ++i) { idx[i].p = probs[i]; idx[i].i = i; }
    qsort(idx, n, sizeof(pi_t), cmp_pi);
    float cum = 0.f;
    int cutoff = n;
    for (int i = 0; i < n; ++i) { cum += idx[i].p; if (cum > p) { cutoff = i + 1; break; } }
    for (int i = cutoff; i < n; ++i) probs[idx[i].i] = 0.f;
    free(idx);
}

/* The running sums q3o_sample compares against top_p (sorted order, topp_cum[0..n)) and against u * total (index order, draw_cum[i], -1
 * where p[i] == 0), and total: lets a test aim u or top_p a few ulps to either side of a boundary, where a sampler that sums in
 * another order would decide differently. */
void q3o_sample_trace(const float* logits, int n, const q3o_sampling* sp, float* topp_cum, float* draw_cum, float* total_out) {
    float* p = (float*)malloc((size_t)n * sizeof(float));
    memcpy(p, logits, (size_t)n * sizeof(float));
    if (sp->temperature > 0.0f && sp->temperature != 1.0f)
        for (int i = 0; i < n; ++i) p[i] /= sp->temperature;
    if (sp->top_k > 0) q3o_top_k_filter(p, n, sp->top_k);
    q3o_softmax(p, n);
    for (int i = 0; i < n; ++i) topp_cum[i] = 0.f;
    if (sp->top_p < 1.0f) {
        pi_t* idx = (pi_t*)malloc((size_t)n * sizeof(pi_t));
        for (int i = 0; i < n; ++i) { idx[i].p = p[i]; idx[i].i = i; }
        qsort(idx, n, sizeof(pi_t), cmp_pi);
        float cum = 0.f;
        for (int i = 0; i < n; ++i) { cum += idx[i].p; topp_cum[i] = cum; }
        free(idx);
        q3o_top_p_filter(p, n, sp->top_p);
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += p[i];
        if (sum > 0.f) for (int i = 0; i < n; ++i) p[i] /= sum;
    }
    float total = 0.f;
    for (int i = 0; i < n; ++i) total += p[i];
    float cum = 0.f;
    for (int i = 0; i < n; ++i) { if (p[i] > 0.f) { cum += p[i]; draw_cum[i] = cum; } else draw_cum[i] = -1.f; }
    *total_out = total;
    free(p);
}

/* :878-905.  NB temperature 0 does NOT mean greedy in the reference (:882 skips the division,
 * sampling proceeds at T=1); greedy is top_k=1.  The final discrete_distribution draw becomes an
 * inverse-CDF walk in index order driven by the supplied uniform u. */
/* margin (optional): how far the decision was from flipping — the smallest of (a) the gap between the top-k threshold and the largest
 * logit below it (after temperature), (b) the distance of the top-p cut's two running sums from top_p, (c) the distance of u * total from
 * the two edges of the drawn element's interval, (d) the relative probability gap between the last element the top-p cut keeps and the
 * first it drops, all relative to a total probability of 1.  A HIP logit that differs from the oracle's
 * in the 5th digit can legitimately change a decision whose margin is smaller than that. */
int64_t q3o_sample_margin(const float* logits, int n, const q3o_sampling* sp, float u, float* margin) {
    float* p = (float*)malloc((size_t)n * sizeof(float));
    memcpy(p, logits, (size_t)n * sizeof(float));
    float mg = INFINITY;
    if (sp->temperature > 0.0f && sp->temperature != 1.0f)
        for (int i = 0; i < n; ++i) p[i] /= sp->temperature;
    if (sp->top_k > 0) {
        if (margin && sp->top_k < n) {
            float* srt = (float*)malloc((size_t)n * sizeof(float));
            memcpy(srt, p, (size_t)n * sizeof(float));
            qsort(srt, n, sizeof(float), cmp_desc);
            float thr = srt[sp->top_k - 1];
            for (int i = sp->top_k; i < n; ++i) if (srt[i] < thr) { if (srt[i] != -INFINITY && thr - srt[i] < mg) mg = thr - srt[i]; break; }
            free(srt);
        }
        q3o_top_k_filter(p, n, sp->top_k);
    }
    q3o_softmax(p, n);
    if (sp->top_p < 1.0f) {
        if (margin) {   /* the running sums on either side of the cut (same order as q3o_top_p_filter) */
            pi_t* idx = (pi_t*)malloc((size_t)n * sizeof(pi_t));
            for (int i = 0; i < n; ++i) { idx[i].p = p[i]; idx[i].i = i; }
            qsort(idx, n, sizeof(pi_t), cmp_pi);
            float cum = 0.f, prev = 0.f;
            for (int i = 0; i < n; ++i) {
                prev = cum; cum += idx[i].p;
                if (cum > sp->top_p) {
                    if (cum - sp->top_p < mg) mg = cum - sp->top_p;
                    if (i > 0 && sp->top_p - prev < mg) mg = sp->top_p - prev;
                    /* (d) WHICH element is the last one kept: the relative gap to the first one dropped (~ their logit gap); if they swap
                     * places the cut keeps a different token, and every running sum behind it in index order moves by a whole probability */
                    if (i + 1 < n && idx[i + 1].p > 0.f) { float gap = (idx[i].p - idx[i + 1].p) / idx[i + 1].p; if (gap < mg) mg = gap; }
                    break;
                }
            }
            free(idx);
        }
        q3o_top_p_filter(p, n, sp->top_p);
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += p[i];
        if (sum > 0.f) for (int i = 0; i < n; ++i) p[i] /= sum;
    }
    float total = 0.f;
    for (int i = 0; i < n; ++i) total += p[i];
    float target = u * total, cum = 0.f, prev = 0.f;
    int64_t pick = -1, last = -1;
    for (int i = 0; i < n; ++i) {
        if (p[i] > 0.f) {
            last = i; prev = cum; cum += p[i];
            if (cum > target) {
                pick = i;
                if (margin && total > 0.f) {
                    if ((cum - target) / total < mg) mg = (cum - target) / total;
                    if (prev > 0.f && (target - prev) / total < mg) mg = (target - prev) / total;
                }
                break;
            }
        }
    }
    free(p);
    if (margin) *margin = mg;
    return pick >= 0 ? pick : last;
}

/* :878-905.  NB temperature 0 does NOT mean greedy in the reference (:882 skips the division,
 * sampling proceeds at T=1); greedy is top_k=1.  The final discrete_distribution draw becomes an
 * inverse-CDF walk in index order driven by the supplied uniform u. */
int64_t q3o_sample(const float* logits, int n, const q3o_sampling* sp, float u) { return q3o_sample_margin(logits, n, sp, u, NULL); }

/* ------------------------------------------------------------------------------------------ */
/* prompt assembly — tts_onnx.cpp:442-539                                                     */
/* ------------------------------------------------------------------------------------------ */

int q3o_build_prompt(q3o_model* m, const int64_t* ids, int n_ids, int lang, const float* speaker, float* prompt, int* S) {
    int H = m->c.hidden;
    /* the reference indexes input_ids[0..3] unguarded (:493, :518): 4 ids is the least it can take; with the usual 5-token frame of
     * an EMPTY text, TTS_EOS lands in the "first text token" slot and the trailing block is just [tts_eos] */
    if (n_ids < 4) FAIL("need at least 4 token ids (role x3 + one more), as the reference indexes input_ids[3]");
    /* 1. tts special embeddings (:459-463) */
    int64_t tts_ids[3] = { TTS_BOS, TTS_EOS, TTS_PAD };
    float* tts = zalloc(3 * (size_t)H);
    if (q3o_text_project(m, tts_ids, 3, tts)) { free(tts); return -1; }
    const float *tts_bos = tts, *tts_eos = tts + H;
    memcpy(m->tts_pad, tts + 2 * H, H * sizeof(float));
    /* 2. codec prefill (:466-476) */
    int64_t cp[8]; int ncp = 0;
    if (lang == 0) { cp[ncp++] = CODEC_NOTHINK; cp[ncp++] = CODEC_THINK_BOS; cp[ncp++] = CODEC_THINK_EOS; }
    else { cp[ncp++] = CODEC_THINK; cp[ncp++] = CODEC_THINK_BOS; cp[ncp++] = LANG_ENGLISH + (lang - 1); cp[ncp++] = CODEC_THINK_EOS; }
    cp[ncp++] = CODEC_PAD; cp[ncp++] = CODEC_BOS;
    float* ce = zalloc((size_t)(ncp + 1) * H);
    if (q3o_codec_embed(m, cp, ncp, ce)) { free(tts); free(ce); return -1; }
    int nrows = ncp;
    if (speaker) { /* speaker row goes in before the last (BOS) row (:481-490) */
        memmove(ce + (size_t)ncp * H, ce + (size_t)(ncp - 1) * H, H * sizeof(float));
        memcpy(ce + (size_t)(ncp - 1) * H, speaker, H * sizeof(float));
        nrows = ncp + 1;
    }
    /* 3. role rows (:493-494) */
    int row = 0;
    if (q3o_text_project(m, ids, 3, prompt)) { free(tts); free(ce); return -1; }
    row = 3;
    /* 4-5. [tts_pad x pad_count, tts_bos] + codec rows (:497-512) */
    int pad_count = ncp - 2 + (speaker ? 1 : 0);
    for (int i = 0; i <= pad_count; ++i, ++row) {
        const float* t = i < pad_count ? m->tts_pad : tts_bos;
        for (int j = 0; j < H; ++j) prompt[(size_t)row * H + j] = t[j] + ce[(size_t)i * H + j];
    }
    /* 6. first text token + codec BOS row (:515-520) */
    int text_start = 3, text_end = n_ids - 2;
    float* ft = zalloc(H);
    if (q3o_text_project(m, ids + text_start, 1, ft)) { free(tts); free(ce); free(ft); return -1; }
    for (int j = 0; j < H; ++j) prompt[(size_t)row * H + j] = ft[j] + ce[(size_t)(pad_count + 1) * H + j];
    ++row;
    (void)nrows;
    /* 8. trailing text rows + tts_eos (:530-536), one text_project call per token */
    free(m->trailing);
    int nt = text_end - (text_start + 1); if (nt < 0) nt = 0;
    m->trailing = zalloc((size_t)(nt + 1) * H);
    for (int i = 0; i < nt; ++i)
        if (q3o_text_project(m, ids + text_start + 1 + i, 1, m->trailing + (size_t)i * H)) { free(tts); free(ce); free(ft); return -1; }
    memcpy(m->trailing + (size_t)nt * H, tts_eos, H * sizeof(float));
    m->trailing_len = nt + 1;
    *S = row;
    free(tts); free(ce); free(ft);
    return 0;
}

int q3o_trailing(q3o_model* m, float* out, int cap_rows, float* pad) {
    int H = m->c.hidden;
    if (pad) memcpy(pad, m->tts_pad, H * sizeof(float));
    int n = m->trailing_len < cap_rows ? m->trailing_len : cap_rows;
    if (out && n > 0) memcpy(out, m->trailing, (size_t)n * H * sizeof(float));
    return m->trailing_len;
}

/* ------------------------------------------------------------------------------------------ */
/* generation loop — tts_onnx.cpp:782-872                                                     */
/* ------------------------------------------------------------------------------------------ */

/* top-1 minus top-2 of a logits row (finite entries only): how far the greedy decision is from flipping */
static float top2_margin(const float* x, int n) {
    float a = -INFINITY, b = -INFINITY;
    for (int i = 0; i < n; ++i) { float v = x[i]; if (v > a) { b = a; a = v; } else if (v > b) b = v; }
    return a - b;
}

/* margins (optional, [max_new_tokens][2 + n_groups]): per generated frame, the top-2 logit margin of the code0 decision (after
 * suppression, before temperature), the smallest top-2 margin over the frame's sub-code decisions, and the SAMPLER decision margin
 * (q3o_sample_margin) of each of the frame's n_groups decisions — the "how close did parity come to flipping" diagnostic of SURVEY.md
 * section 7. */
/* debugging aid for the parity tests: the logits row of ONE decision (frame, group) of the next q3o_generate* call */
static int g_dump_frame = -1, g_dump_group = -1;
static float* g_dump_buf = NULL;
void q3o_set_logits_dump(int frame, int group, float* buf) { g_dump_frame = frame; g_dump_group = group; g_dump_buf = buf; }

static int generate_impl(q3o_model* m, const float* prompt, int S, const q3o_sampling* sp, uint64_t seed, uint32_t stream,
                         int cp_cached, int ignore_eos, int64_t* codes, float* margins) {
    const q3o_config* c = &m->c;
    int H = c->hidden, V = c->vocab, G = c->n_groups, SV = c->sub_vocab;
    float* logits_all = zalloc((size_t)S * V);
    float* last = zalloc(V);
    float* sub_logits = zalloc(SV);
    float* seq = zalloc((size_t)(G + 1) * H);
    float* x = zalloc(H);
    float* e = zalloc(H);
    if (q3o_prefill(m, prompt, S, logits_all, NULL)) { free(logits_all); free(last); free(sub_logits); free(seq); free(x); free(e); return -1; }
    memcpy(last, logits_all + (size_t)(S - 1) * V, V * sizeof(float)); /* :797-798 */
    int F = 0;
    for (int step = 0; step < sp->max_new_tokens; ++step) {
        /* suppress 2048..3071 except EOS (:803-807); benchmark mode also suppresses EOS */
        for (int i = c->suppress_begin; i < c->suppress_end; ++i)
            if (i != c->codec_eos || ignore_eos) last[i] = -INFINITY;
        float dm = INFINITY, dm1 = INFINITY;
        if (g_dump_buf && F == g_dump_frame && g_dump_group == 0) memcpy(g_dump_buf, last, (size_t)V * sizeof(float));
        int64_t code0 = q3o_sample_margin(last, V, sp, q3o_rng_uniform(seed, stream, (uint32_t)step, 0), margins ? &dm : NULL); /* :810 */
        if (code0 == c->codec_eos) break;                                                          /* :812 */
        if (margins) { margins[(size_t)(2 + G) * F] = top2_margin(last, V); margins[(size_t)(2 + G) * F + 1] = INFINITY; margins[(size_t)(2 + G) * F + 2] = dm; }
        /* predict_subcodes (:851-872): seq = [last_hidden, codec_embed(code0), sub embeds...] */
        int64_t* frame = codes + (size_t)F * G;
        frame[0] = code0;
        memcpy(seq, m->last_hidden, H * sizeof(float));
        q3o_codec_embed(m, &code0, 1, seq + H);
        for (int j = 0; j < G - 1; ++j) {
            if (!cp_cached) q3o_code_predictor(m, seq, j + 2, j, sub_logits);                      /* :863 */
            else if (j == 0) cp_cached_step(m, seq, 0, 2, 0, sub_logits);
            else cp_cached_step(m, seq + (size_t)(j + 1) * H, j + 1, j + 2, j, sub_logits);
            if (g_dump_buf && F == g_dump_frame && g_dump_group == j + 1) memcpy(g_dump_buf, sub_logits, (size_t)SV * sizeof(float));
            int64_t sc = q3o_sample_margin(sub_logits, SV, sp, q3o_rng_uniform(seed, stream, (uint32_t)step, (uint32_t)(j + 1)), margins ? &dm1 : NULL); /* :864 */
            frame[j + 1] = sc;
            if (margins) {
                float mg = top2_margin(sub_logits, SV);
                if (mg < margins[(size_t)(2 + G) * F + 1]) margins[(size_t)(2 + G) * F + 1] = mg;
                margins[(size_t)(2 + G) * F + 3 + j] = dm1;
            }
            q3o_cp_embed(m, sc, j, seq + (size_t)(j + 2) * H);                                     /* :867-868 */
        }
        ++F;
        /* next input = codec_embed(code0) + sum_i cp_embed(sub_i, i) (+ text row | tts_pad), fp32,
         * in that order (:824-842) */
        q3o_codec_embed(m, &code0, 1, x);
        for (int i = 0; i < G - 1; ++i) {
            q3o_cp_embed(m, frame[i + 1], i, e);
            for (int j = 0; j < H; ++j) x[j] += e[j];
        }
        const float* t = step < m->trailing_len ? m->trailing + (size_t)step * H : m->tts_pad;
        for (int j = 0; j < H; ++j) x[j] += t[j];
        if (q3o_decode(m, x, last, NULL)) break;                                                   /* :845 */
    }
    free(logits_all); free(last); free(sub_logits); free(seq); free(x); free(e);
    return F;
}

int q3o_generate(q3o_model* m, const float* prompt, int S, const q3o_sampling* sp, uint64_t seed, uint32_t stream,
                 int cp_cached, int ignore_eos, int64_t* codes) {
    return generate_impl(m, prompt, S, sp, seed, stream, cp_cached, ignore_eos, codes, NULL);
}
int q3o_generate_margins(q3o_model* m, const float* prompt, int S, const q3o_sampling* sp, uint64_t seed, uint32_t stream,
                         int cp_cached, int ignore_eos, int64_t* codes, float* margins) {
    return generate_impl(m, prompt, S, sp, seed, stream, cp_cached, ignore_eos, codes, margins);
}

/* ------------------------------------------------------------------------------------------ */
/* 12 Hz codec decoder ("tokenizer12hz_decode", tts_onnx.cpp:759-776).  [HINT] architecture:   */
/* transformers Qwen3OmniMoeCode2Wav, modeling_qwen3_omni_moe.py:3180-3266, 3542-3697.         */
/* Activations are time-major [T][C].                                                          */
/* ------------------------------------------------------------------------------------------ */

static void snake(const float* x, int T, int C, const snake_w* s, float* y) {
    float* ea = (float*)malloc((size_t)C * sizeof(float));
    float* ib = (float*)malloc((size_t)C * sizeof(float));
    for (int c = 0; c < C; ++c) { ea[c] = expf(s->alpha[c]); ib[c] = 1.0f / (expf(s->beta[c]) + 0.000000001f); }
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c) {
            float v = x[(size_t)t * C + c], sn = sinf(v * ea[c]);
            y[(size_t)t * C + c] = v + ib[c] * (sn * sn);
        }
    free(ea); free(ib);
}

/* causal Conv1d, stride 1: left pad (k-1)*dil (CausalConvNet, modeling :3180-3213) */
static void cconv(const float* x, int T, const conv_w* c, int dil, float* y) {
    int cin = c->cin, cout = c->cout, k = c->k;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t)
        for (int co = 0; co < cout; ++co) {
            float acc = c->b[co];
            for (int tap = 0; tap < k; ++tap) {
                int ts = t - (k - 1 - tap) * dil;
                if (ts < 0) continue;
                acc += dotf(x + (size_t)ts * cin, c->wt + ((size_t)tap * cout + co) * cin, cin);
            }
            y[(size_t)t * cout + co] = acc;
        }
}

static int tconv_len(const q3o_config* cfg, int T, int k, int s) {
    int pad = k - s;
    int left = cfg->cd_tconv_trim == 0 ? pad : 0, right = pad;
    return (T - 1) * s + k - left - right;
}
/* causal ConvTranspose1d (CausalTransConvNet, modeling :3216-3228) */
static void tconv(const q3o_config* cfg, const float* x, int T, const conv_w* c, int s, float* y) {
    int cin = c->cin, cout = c->cout, k = c->k;
    int left = cfg->cd_tconv_trim == 0 ? k - s : 0;
    int To = tconv_len(cfg, T, k, s);
#pragma omp parallel for schedule(static)
    for (int jo = 0; jo < To; ++jo) {
        int j = jo + left;
        for (int co = 0; co < cout; ++co) {
            float acc = c->b[co];
            for (int t = j / s; t >= 0 && j - t * s < k; --t) {
                if (t >= T) continue;
                int tap = j - t * s;
                acc += dotf(x + (size_t)t * cin, c->wt + ((size_t)tap * cout + co) * cin, cin);
            }
            y[(size_t)jo * cout + co] = acc;
        }
    }
}

static void convnext(const upstage_w* u, float* x, int T, int C) {
    float* h = zalloc((size_t)T * C);
    float* a = zalloc((size_t)T * 4 * C);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t) {
        float* hr = h + (size_t)t * C;
        for (int c = 0; c < C; ++c) { /* depthwise causal k7 */
            float acc = u->dw.b[c];
            for (int tap = 0; tap < 7; ++tap) { int ts = t - (6 - tap); if (ts >= 0) acc += u->dw.w[c * 7 + tap] * x[(size_t)ts * C + c]; }
            hr[c] = acc;
        }
        float mean = 0.f, var = 0.f; /* LayerNorm eps 1e-6 */
        for (int c = 0; c < C; ++c) mean += hr[c];
        mean /= (float)C;
        for (int c = 0; c < C; ++c) { float d = hr[c] - mean; var += d * d; }
        var /= (float)C;
        float r = 1.0f / sqrtf(var + 1e-6f);
        for (int c = 0; c < C; ++c) hr[c] = (hr[c] - mean) * r * u->ln_w[c] + u->ln_b[c];
    }
    linear(h, T, C, u->pw1_w, u->pw1_b, 4 * C, a, 4 * C);
    for (size_t i = 0; i < (size_t)T * 4 * C; ++i) a[i] = geluf(a[i]);
    linear(a, T, 4 * C, u->pw2_w, u->pw2_b, C, h, C);
    for (size_t i = 0; i < (size_t)T * C; ++i) x[i] += u->gamma[i % C] * h[i];
    free(h); free(a);
}

int64_t q3o_vocoder_len(const q3o_config* c, int F) {
    int64_t T = F;
    for (int s = 0; s < c->cd_n_up; ++s) T *= c->cd_up_ratios[s];
    for (int i = 0; i < c->cd_n_blocks; ++i) T = tconv_len(c, (int)T, 2 * c->cd_up_rates[i], c->cd_up_rates[i]);
    return T;
}

/* stage < 0: run to PCM */
static int64_t vocoder_run(q3o_model* m, const int64_t* codes, int F, int stage, float* out, int64_t cap) {
    const q3o_config* c = &m->c;
    int CH = c->cd_hidden, G = c->n_groups;
    if (F < 1) { snprintf(g_err, sizeof g_err, "vocoder: F < 1"); return -1; }
    float* h = zalloc((size_t)F * CH);
    /* code_embedding(codes + offset).mean(1) (modeling :3675) */
    for (int t = 0; t < F; ++t) {
        for (int g = 0; g < G; ++g) {
            int64_t id = codes[(size_t)t * G + g];
            if (id < 0 || id >= c->cd_codebook) { free(h); snprintf(g_err, sizeof g_err, "vocoder: code %lld out of range", (long long)id); return -1; }
            const float* r = m->code_embed + ((size_t)g * c->cd_codebook + id) * CH;
            for (int j = 0; j < CH; ++j) h[(size_t)t * CH + j] += r[j];
        }
        for (int j = 0; j < CH; ++j) h[(size_t)t * CH + j] /= (float)G;
    }
    dec_dims D = cd_dims(c);
    size_t kvn = (size_t)c->cd_layers * c->cd_heads * F * c->cd_head_dim;
    float *kc = zalloc(kvn), *vc = zalloc(kvn);
    dec_forward(&D, m->cl, kc, vc, F, h, F, 0);
    free(kc); free(vc);
    for (int t = 0; t < F; ++t) rmsnorm(h + (size_t)t * CH, m->c_norm, CH, c->cd_rms_eps, h + (size_t)t * CH);
    int T = F, C = CH;
    float* cur = h;
#define TAP(st) if (stage == (st)) { int64_t n = (int64_t)T * C; if (n > cap) n = cap; memcpy(out, cur, (size_t)n * sizeof(float)); free(cur); return (int64_t)T * C; }
    TAP(0)
    for (int s = 0; s < c->cd_n_up; ++s) {
        int f = c->cd_up_ratios[s];
        int To = tconv_len(c, T, f, f);
        float* y = zalloc((size_t)To * C);
        tconv(c, cur, T, &m->up[s].tconv, f, y);
        free(cur); cur = y; T = To;
        convnext(&m->up[s], cur, T, C);
    }
    TAP(1)
    {
        int Dd = c->cd_decoder_dim;
        float* y = zalloc((size_t)T * Dd);
        cconv(cur, T, &m->conv_in, 1, y);
        free(cur); cur = y; C = Dd;
    }
    TAP(2)
    static const int dil[3] = { 1, 3, 9 };
    for (int i = 0; i < c->cd_n_blocks; ++i) {
        block_w* B = &m->blk[i];
        int r = c->cd_up_rates[i], Co = C / 2;
        float* a = zalloc((size_t)T * C);
        snake(cur, T, C, &B->act, a);
        int To = tconv_len(c, T, 2 * r, r);
        float* y = zalloc((size_t)To * Co);
        tconv(c, a, T, &B->tconv, r, y);
        free(a); free(cur); cur = y; T = To; C = Co;
        float* t1 = zalloc((size_t)T * C);
        float* t2 = zalloc((size_t)T * C);
        for (int u = 0; u < 3; ++u) {
            snake(cur, T, C, &B->res[u].a1, t1);
            cconv(t1, T, &B->res[u].c1, dil[u], t2);
            snake(t2, T, C, &B->res[u].a2, t1);
            cconv(t1, T, &B->res[u].c2, 1, t2);
            for (size_t j = 0; j < (size_t)T * C; ++j) cur[j] = t2[j] + cur[j];
        }
        free(t1); free(t2);
        TAP(3 + i)
    }
#undef TAP
    float* a = zalloc((size_t)T * C);
    snake(cur, T, C, &m->snake_out, a);
    float* w = zalloc(T);
    cconv(a, T, &m->conv_out, 1, w);
    free(a); free(cur);
    int64_t n = T < cap ? T : cap;
    for (int64_t i = 0; i < n; ++i) out[i] = w[i] < -1.f ? -1.f : (w[i] > 1.f ? 1.f : w[i]); /* clamp (modeling :3684) */
    free(w);
    return T;
}

int64_t q3o_vocoder(q3o_model* m, const int64_t* codes, int F, float* pcm, int64_t cap) { return vocoder_run(m, codes, F, -1, pcm, cap); }
int64_t q3o_vocoder_tap(q3o_model* m, const int64_t* codes, int F, int stage, float* out, int64_t cap) { return vocoder_run(m, codes, F, stage, out, cap); }

/* ------------------------------------------------------------------------------------------ */
/* speaker encoder — run_speaker_encoder (src/tts_onnx.cpp:367-403).  The reference runs an opaque   */
/* speaker_encoder.onnx; the network restated here is the published ECAPA-TDNN of the Qwen audio    */
/* stack [HINT: transformers models/qwen2_5_omni/modeling_qwen2_5_omni.py:2412-2716], pinned on     */
/* tests/golden/hf_speaker.npz.  Activations are channel-major [C][T]; every Conv1d uses "same"     */
/* reflect padding (so frames must exceed the largest pad, 4).                                      */
/* ------------------------------------------------------------------------------------------ */
static inline int reflect_idx(int i, int T) { return i < 0 ? -i : (i >= T ? 2 * (T - 1) - i : i); }

/* y[co][t] = act(b[co] + sum_{ci,j} W[co][ci][j] * (x[ci] (+ x2[ci]))[reflect(t + (j - k/2) * dil)]); act: 0 none, 1 relu */
static void conv1d_reflect(const float* x, const float* x2, int T, const conv_w* c, int dil, int act, float* y) {
    const int k = c->k, half = k / 2;
#pragma omp parallel for schedule(static) if (c->cout * T > 4096)
    for (int co = 0; co < c->cout; ++co) {
        for (int t = 0; t < T; ++t) {
            float acc = c->b[co];
            for (int ci = 0; ci < c->cin; ++ci) {
                const float* w = c->w + ((size_t)co * c->cin + ci) * k;
                const float* xr = x + (size_t)ci * T;
                const float* xr2 = x2 ? x2 + (size_t)ci * T : NULL;
                for (int j = 0; j < k; ++j) {
                    const int src = reflect_idx(t + (j - half) * dil, T);
                    const float v = xr2 ? xr[src] + xr2[src] : xr[src];
                    acc = fmaf(w[j], v, acc);
                }
            }
            y[(size_t)co * T + t] = act == 1 ? (acc > 0.f ? acc : 0.f) : acc;
        }
    }
}

int q3o_speaker_encoder(q3o_model* m, const float* mel, int T, float* out) {
    const q3o_config* c = &m->c;
    if (c->spk_enc_dim <= 0) FAIL("model has no speaker encoder");
    if (T < 5) FAIL("speaker encoder needs at least 5 mel frames (reflect padding), got %d", T);
    const int SC = c->spk_channels, sub = SC / c->spk_scale, C3 = 3 * SC;
    float* h = zalloc((size_t)SC * T);       /* block input / residual */
    float* a = zalloc((size_t)SC * T);
    float* r2 = zalloc((size_t)SC * T);
    float* cc = zalloc((size_t)SC * T);
    float* cat = zalloc((size_t)C3 * T);
    conv1d_reflect(mel, NULL, T, &m->s_tdnn0, 1, 1, h);
    for (int i = 0; i < 3; ++i) {
        conv1d_reflect(h, NULL, T, &m->s_tdnn1[i], 1, 1, a);
        /* Res2Net: chunk 0 passes through, chunk 1 = f(chunk 1), chunk j = f(chunk j + out j-1) */
        memcpy(r2, a, (size_t)sub * T * sizeof(float));
        for (int j = 1; j < c->spk_scale; ++j)
            conv1d_reflect(a + (size_t)j * sub * T, j >= 2 ? r2 + (size_t)(j - 1) * sub * T : NULL, T, &m->s_res[i][j - 1], i + 2, 1,
                           r2 + (size_t)j * sub * T);
        conv1d_reflect(r2, NULL, T, &m->s_tdnn2[i], 1, 1, cc);
        /* squeeze-excitation: per-channel gate from the time mean */
        float* mean = zalloc(SC); float* s1 = zalloc(c->spk_se); float* g = zalloc(SC);
        for (int ch = 0; ch < SC; ++ch) { float s = 0.f; for (int t = 0; t < T; ++t) s += cc[(size_t)ch * T + t]; mean[ch] = s / (float)T; }
        conv1d_reflect(mean, NULL, 1, &m->s_se1[i], 1, 1, s1);
        conv1d_reflect(s1, NULL, 1, &m->s_se2[i], 1, 0, g);
        for (int ch = 0; ch < SC; ++ch) {
            const float gate = 1.0f / (1.0f + expf(-g[ch]));
            for (int t = 0; t < T; ++t) {
                const float v = cc[(size_t)ch * T + t] * gate + h[(size_t)ch * T + t];
                h[(size_t)ch * T + t] = v;
                cat[((size_t)i * SC + ch) * T + t] = v;
            }
        }
        free(mean); free(s1); free(g);
    }
    float* mf = zalloc((size_t)C3 * T);
    conv1d_reflect(cat, NULL, T, &m->s_mfa, 1, 1, mf);
    /* attentive statistics pooling */
    float* att_in = zalloc((size_t)3 * C3 * T);
    memcpy(att_in, mf, (size_t)C3 * T * sizeof(float));
    for (int ch = 0; ch < C3; ++ch) {
        const float* x = mf + (size_t)ch * T;
        float mu = 0.f;
        for (int t = 0; t < T; ++t) mu += x[t] / (float)T;   /* sum of (1/T)*x, as the masked mean is written */
        float var = 0.f;
        for (int t = 0; t < T; ++t) { const float d = x[t] - mu; var += d * d / (float)T; }
        const float sd = sqrtf(var > 1e-12f ? var : 1e-12f);
        for (int t = 0; t < T; ++t) { att_in[((size_t)C3 + ch) * T + t] = mu; att_in[((size_t)2 * C3 + ch) * T + t] = sd; }
    }
    float* at = zalloc((size_t)c->spk_att * T);
    conv1d_reflect(att_in, NULL, T, &m->s_asp_tdnn, 1, 1, at);
    for (size_t i = 0; i < (size_t)c->spk_att * T; ++i) at[i] = tanhf(at[i]);
    float* w = zalloc((size_t)C3 * T);
    conv1d_reflect(at, NULL, T, &m->s_asp_conv, 1, 0, w);
    float* pooled = zalloc((size_t)2 * C3);
    for (int ch = 0; ch < C3; ++ch) {
        float* wr = w + (size_t)ch * T;
        const float* x = mf + (size_t)ch * T;
        softmax_libm(wr, T);
        float mu = 0.f;
        for (int t = 0; t < T; ++t) mu += wr[t] * x[t];
        float var = 0.f;
        for (int t = 0; t < T; ++t) { const float d = x[t] - mu; var += wr[t] * d * d; }
        pooled[ch] = mu;
        pooled[C3 + ch] = sqrtf(var > 1e-12f ? var : 1e-12f);
    }
    conv1d_reflect(pooled, NULL, 1, &m->s_fc, 1, 0, out);
    free(h); free(a); free(r2); free(cc); free(cat); free(mf); free(att_in); free(at); free(w); free(pooled);
    return 0;
}


/* synthesize_tokens (tts_onnx.cpp:405-436) */
int64_t q3o_synthesize_tokens(q3o_model* m, const int64_t* ids, int n_ids, int lang, const q3o_sampling* sp,
                              uint64_t seed, uint32_t stream, float* pcm, int64_t cap, int64_t* codes, int* n_frames) {
    int H = m->c.hidden, S = 0;
    float* prompt = zalloc((size_t)16 * H);
    if (q3o_build_prompt(m, ids, n_ids, lang, NULL, prompt, &S)) { free(prompt); return -1; }
    int64_t* cb = codes ? codes : (int64_t*)malloc((size_t)sp->max_new_tokens * m->c.n_groups * sizeof(int64_t));
    int F = q3o_generate(m, prompt, S, sp, seed, stream, 1, 0, cb);
    free(prompt);
    if (n_frames) *n_frames = F;
    int64_t n = 0;
    if (F > 0) n = q3o_vocoder(m, cb, F, pcm, cap); /* empty result when no frames (:418) */
    if (!codes) free(cb);
    return F < 0 ? -1 : n;
}
