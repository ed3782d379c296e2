// C ABI of the video Q-former path (include/infv_vqf.h): workspace + launch sequence of one chunk.
#include "../../include/infv_vqf.h"
#include "capi_common.h"
#include "vqf_internal.h"

#include <new>

using namespace infv;

struct infv_vqf_s {
    infv_vqf_config cfg;
    int dev = 0;
    // workspaces (grown on demand; a growing call synchronises the device first)
    DeviceBuf part, h_a, h_b, h1, h2, qkv, sa, xq, along, qt, S, O, merged, inter, kbar;
};

namespace {

int check_cfg(const infv_vqf_config& c) {
    if (c.n_layers < 1 || c.n_layers > INFV_VQF_MAX_LAYERS) return fail(INFV_ERR_INVALID, "n_layers must be 1..%d", INFV_VQF_MAX_LAYERS);
    if (c.n_heads < 1 || c.hidden != c.n_heads * 64) return fail(INFV_ERR_UNSUPPORTED, "hidden must be n_heads * 64");
    if (c.n_query < 1 || c.n_query > 32) return fail(INFV_ERR_UNSUPPORTED, "n_query must be 1..32");
    if (c.hidden % 64 || c.inter % 64 || c.enc_width % 32 || (c.proj_out % 64)) return fail(INFV_ERR_UNSUPPORTED, "widths must be multiples of 64 (enc_width: 32)");
    if (c.hidden > 4096 || c.inter > 4096 || c.proj_out > 4096 || 3 * c.hidden > 4096) return fail(INFV_ERR_UNSUPPORTED, "row widths above 4096 are not supported");
    if (c.nb_samples < 1) return fail(INFV_ERR_INVALID, "nb_samples must be >= 1");
    if (c.tokens_per_frame < 1 || c.tokens_per_frame % 32) return fail(INFV_ERR_UNSUPPORTED, "tokens_per_frame must be a multiple of 32");
    return INFV_OK;
}

// y = epilogue( x [M][K] . W^T ) with up to 3 stacked weight matrices of n_out rows each
struct LinearCall {
    const float* x; int M, K;
    const infv_linear* lin[kQfMaxSeg]; int n_lin; int n_out;      // output width = n_lin * n_out
    int act = QF_ACT_NONE;
    const float* residual = nullptr; int res_rows = 1;
    const infv_layernorm* ln = nullptr;
    float* y;
};

int run_linear(infv_vqf_s* h, const LinearCall& c, hipStream_t stream) {
    const int width = c.n_lin * c.n_out;
    const int sk = qf_pick_splitk(c.M, width, c.K, 1);
    const size_t need = (size_t)sk * c.M * width * sizeof(float);
    if (need > h->part.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->part.reserve(need)); }
    QfGemm g{};
    g.A = c.x; g.lda = c.K; g.strideA = 0;
    for (int i = 0; i < c.n_lin; ++i) g.B[i] = c.lin[i]->w;
    g.ldb = c.K; g.strideB = 0; g.seg_rows = c.n_out;
    g.C = h->part.as<float>(); g.ldc = width; g.strideC = 0; g.split_stride = (long)c.M * width;
    g.M = c.M; g.N = width; g.k_per_split = c.K / sk; g.splitk = sk; g.nbatch = 1;
    HIP_TRY(launch_qf_gemm(g, false, stream));
    QfEpilogue e{};
    e.parts = h->part.as<float>(); e.nsplit = sk; e.split_stride = g.split_stride; e.ld_in = width;
    for (int i = 0; i < c.n_lin; ++i) e.bias[i] = c.lin[i]->b;
    e.seg_cols = c.n_out; e.act = c.act; e.scale = 1.f; e.res_scale = 1.f;
    e.residual = c.residual; e.ld_res = width; e.res_rows = c.res_rows;
    e.gamma = c.ln ? c.ln->gamma : nullptr; e.beta = c.ln ? c.ln->beta : nullptr; e.eps = h->cfg.ln_eps;
    e.out = c.y; e.ld_out = width; e.M = c.M; e.width = width;
    HIP_TRY(launch_qf_epilogue(e, stream));
    return INFV_OK;
}

// frames [nb][n_tokens][d], xq [nb*Q][hidden] -> merged [nb*Q][hidden]
int short_attention(infv_vqf_s* h, const float* frames, int nb, int n_tokens, const float* xq, const infv_linear* key,
                    const infv_linear* value, const float* along, float* merged, hipStream_t stream) {
    const infv_vqf_config& c = h->cfg;
    const int Q = c.n_query, H = c.n_heads, d = c.enc_width, rows = H * Q;
    if (n_tokens < 32 || n_tokens % 32) return fail(INFV_ERR_INVALID, "n_tokens must be a positive multiple of 32");
    const int sk = qf_pick_splitk(rows, d, n_tokens, nb);
    const size_t needS = (size_t)nb * rows * n_tokens * sizeof(float);
    const size_t needO = (size_t)sk * nb * rows * d * sizeof(float);
    const size_t needQt = (size_t)nb * rows * d * sizeof(float);
    if (needS > h->S.bytes || needO > h->O.bytes || needQt > h->qt.bytes) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(h->S.reserve(needS)); HIP_TRY(h->O.reserve(needO)); HIP_TRY(h->qt.reserve(needQt));
    }
    HIP_TRY(launch_qf_qtilde(xq, nb, Q, H, d, key->w, h->qt.as<float>(), stream));
    QfGemm g{};                                             // S[b] = qt[b] . frames[b]^T
    g.A = h->qt.as<float>(); g.lda = d; g.strideA = (long)rows * d;
    g.B[0] = frames; g.ldb = d; g.strideB = (long)n_tokens * d; g.seg_rows = n_tokens;
    g.C = h->S.as<float>(); g.ldc = n_tokens; g.strideC = (long)rows * n_tokens; g.split_stride = 0;
    g.M = rows; g.N = n_tokens; g.k_per_split = d; g.splitk = 1; g.nbatch = nb;
    HIP_TRY(launch_qf_gemm(g, false, stream));
    HIP_TRY(launch_qf_softmax_rows(h->S.as<float>(), (long)nb * rows, n_tokens, n_tokens, stream));
    QfGemm p{};                                             // O[b] = P[b] . frames[b]
    p.A = h->S.as<float>(); p.lda = n_tokens; p.strideA = (long)rows * n_tokens;
    p.B[0] = frames; p.ldb = d; p.strideB = (long)n_tokens * d; p.seg_rows = d;
    p.C = h->O.as<float>(); p.ldc = d; p.strideC = (long)rows * d; p.split_stride = (long)nb * rows * d;
    p.M = rows; p.N = d; p.k_per_split = n_tokens / sk; p.splitk = sk; p.nbatch = nb;
    HIP_TRY(launch_qf_gemm(p, true, stream));
    // per-head value projection as a batched GEMM over (chunk, head): [Q x d] . Wv_h^T -> [Q x 64], then
    // bias + merge with the long-term context in the row epilogue (Qformer.py:298-304)
    HIP_TRY(launch_qf_sum_slabs(h->O.as<float>(), sk, p.split_stride, (long)nb * rows * d, stream));
    const int M2 = nb * Q, hidden = c.hidden;
    int sk2 = 8;
    while (sk2 > 1 && d % (32 * sk2)) sk2 >>= 1;
    const size_t needP = (size_t)sk2 * M2 * hidden * sizeof(float);
    if (needP > h->part.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->part.reserve(needP)); }
    for (int b = 0; b < nb; ++b) {
        QfGemm v{};
        v.A = h->O.as<float>() + (long)b * rows * d; v.lda = d; v.strideA = (long)Q * d;          // batch = head
        v.B[0] = value->w; v.ldb = d; v.strideB = 64L * d; v.seg_rows = 64;
        v.C = h->part.as<float>() + (long)b * Q * hidden; v.ldc = hidden; v.strideC = 64;
        v.split_stride = (long)M2 * hidden;
        v.M = Q; v.N = 64; v.k_per_split = d / sk2; v.splitk = sk2; v.nbatch = H;
        HIP_TRY(launch_qf_gemm(v, false, stream));
    }
    QfEpilogue e{};
    e.parts = h->part.as<float>(); e.nsplit = sk2; e.split_stride = (long)M2 * hidden; e.ld_in = hidden;
    e.bias[0] = value->b; e.seg_cols = hidden; e.act = QF_ACT_NONE;
    e.scale = along ? c.alpha : 1.f; e.res_scale = (float)(1.0 - (double)c.alpha);
    e.residual = along; e.ld_res = hidden; e.res_rows = M2;
    e.out = merged; e.ld_out = hidden; e.M = M2; e.width = hidden; e.eps = c.ln_eps;
    HIP_TRY(launch_qf_epilogue(e, stream));
    return INFV_OK;
}

}  // namespace

extern "C" {

int infv_vqf_create(const infv_vqf_config* cfg, infv_vqf_handle* out) {
    if (!cfg || !out) return fail(INFV_ERR_INVALID, "null argument");
    if (int rc = check_cfg(*cfg)) return rc;
    infv_vqf_s* h = new (std::nothrow) infv_vqf_s();
    if (!h) return fail(INFV_ERR_INVALID, "out of host memory");
    h->cfg = *cfg;
    HIP_TRY(hipGetDevice(&h->dev));
    const infv_vqf_config& c = *cfg;
    const size_t row = (size_t)c.n_query * sizeof(float);
    hipError_t e = hipSuccess;
    auto rs = [&](DeviceBuf& b, size_t n) { if (e == hipSuccess) e = b.reserve(n); };
    rs(h->h_a, row * c.hidden); rs(h->h_b, row * c.hidden); rs(h->h1, row * c.hidden); rs(h->h2, row * c.hidden);
    rs(h->qkv, row * 3 * c.hidden); rs(h->sa, row * c.hidden); rs(h->xq, row * c.hidden); rs(h->along, row * c.hidden);
    rs(h->merged, row * c.hidden); rs(h->inter, row * c.inter);
    if (e != hipSuccess) { delete h; return fail(INFV_ERR_HIP, "workspace allocation failed: %s", hipGetErrorString(e)); }
    *out = h;
    return INFV_OK;
}

int infv_vqf_destroy(infv_vqf_handle h) {
    if (!h) return INFV_OK;
    (void)hipDeviceSynchronize();
    delete h;
    return INFV_OK;
}

int infv_vqf_short_attention(infv_vqf_handle h, const float* frames, int32_t n_tokens, const float* xq,
                             const infv_linear* key, const infv_linear* value, const float* a_long,
                             float* merged, void* stream) {
    if (!h || !frames || !xq || !key || !value || !merged || !key->w || !value->w || !value->b)
        return fail(INFV_ERR_INVALID, "null argument");
    return short_attention(h, frames, 1, n_tokens, xq, key, value, a_long, merged, static_cast<hipStream_t>(stream));
}

int infv_vqf_encode_chunk(infv_vqf_handle h, const infv_ltm_handle* ltm, const float* frames, int32_t T,
                          const infv_vqf_weights* w, const double* u, int32_t new_video,
                          float* hidden_out, float* llama_out, void* stream_) {
    if (!h || !frames || !w) return fail(INFV_ERR_INVALID, "null argument");
    const infv_vqf_config& c = h->cfg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool use_ltm = c.alpha != 1.0f;                      // Qformer.py:220-223
    if (use_ltm && !ltm) return fail(INFV_ERR_INVALID, "alpha != 1 needs the per-layer LTM handles");
    if (T < 1) return fail(INFV_ERR_INVALID, "T must be >= 1");
    if (llama_out && (c.proj_out <= 0 || !w->llama_proj.w)) return fail(INFV_ERR_INVALID, "llama_out without llama_proj");
    const int Q = c.n_query, Hd = c.hidden, n_tokens = T * c.tokens_per_frame;

    // embeddings: LayerNorm of the learned query tokens (Qformer.py:108-112)
    QfEpilogue e{};
    e.parts = w->query_tokens; e.nsplit = 1; e.split_stride = 0; e.ld_in = Hd; e.seg_cols = Hd;
    e.gamma = w->emb_ln.gamma; e.beta = w->emb_ln.beta; e.eps = c.ln_eps; e.scale = 1.f; e.res_scale = 1.f;
    e.out = h->h_a.as<float>(); e.ld_out = Hd; e.M = Q; e.width = Hd; e.res_rows = 1;
    HIP_TRY(launch_qf_epilogue(e, stream));
    float* hcur = h->h_a.as<float>();
    float* hnext = h->h_b.as<float>();

    if (use_ltm) {
        const size_t need = (size_t)T * c.enc_width * sizeof(float);
        if (need > h->kbar.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->kbar.reserve(need)); }
        if (new_video)
            for (int l = 0; l < c.n_layers; ++l)
                if (int rc = infv_ltm_reset(ltm[l])) return rc;
        if (int rc = infv_ltm_pool(ltm[0], frames, T, h->kbar.as<float>(), stream_)) return rc;   // shared by all layers
    }

    for (int l = 0; l < c.n_layers; ++l) {
        const infv_vqf_layer& L = w->layer[l];
        // ---- self-attention over the query tokens + output (Qformer.py:442-470 -> BertAttention)
        LinearCall qkv{hcur, Q, Hd, {&L.self_q, &L.self_k, &L.self_v}, 3, Hd};
        qkv.y = h->qkv.as<float>();
        if (int rc = run_linear(h, qkv, stream)) return rc;
        HIP_TRY(launch_qf_self_attention(h->qkv.as<float>(), 1, Q, c.n_heads, h->sa.as<float>(), stream));
        LinearCall so{h->sa.as<float>(), Q, Hd, {&L.self_o}, 1, Hd};
        so.residual = hcur; so.res_rows = Q; so.ln = &L.self_ln; so.y = h->h1.as<float>();
        if (int rc = run_linear(h, so, stream)) return rc;
        // ---- cross-attention: query, long-term memory, short-term attention, merge, output
        LinearCall xq{h->h1.as<float>(), Q, Hd, {&L.x_q}, 1, Hd};
        xq.y = h->xq.as<float>();
        if (int rc = run_linear(h, xq, stream)) return rc;
        const float* along = nullptr;
        if (use_ltm) {
            infv_ltm_proj pr{};
            pr.wk = L.x_k.w; pr.bk = L.x_k.b; pr.wv = L.x_v.w; pr.bv = L.x_v.b;
            const double* ul = u ? u + (size_t)l * c.nb_samples : nullptr;
            if (int rc = infv_ltm_step(ltm[l], h->kbar.as<float>(), T, h->xq.as<float>(), Q, &pr, ul,
                                       h->along.as<float>(), stream_)) return rc;
            along = h->along.as<float>();
        }
        if (int rc = short_attention(h, frames, 1, n_tokens, h->xq.as<float>(), &L.x_k, &L.x_v, along,
                                     h->merged.as<float>(), stream)) return rc;
        LinearCall xo{h->merged.as<float>(), Q, Hd, {&L.x_o}, 1, Hd};
        xo.residual = h->h1.as<float>(); xo.res_rows = Q; xo.ln = &L.x_ln; xo.y = h->h2.as<float>();
        if (int rc = run_linear(h, xo, stream)) return rc;
        // ---- query FFN (Qformer.py:519-522)
        LinearCall fi{h->h2.as<float>(), Q, Hd, {&L.ffn_in}, 1, c.inter};
        fi.act = QF_ACT_GELU; fi.y = h->inter.as<float>();
        if (int rc = run_linear(h, fi, stream)) return rc;
        LinearCall fo{h->inter.as<float>(), Q, c.inter, {&L.ffn_out}, 1, Hd};
        fo.residual = h->h2.as<float>(); fo.res_rows = Q; fo.ln = &L.ffn_ln; fo.y = hnext;
        if (int rc = run_linear(h, fo, stream)) return rc;
        float* t = hcur; hcur = hnext; hnext = t;
    }
    if (hidden_out)
        HIP_TRY(hipMemcpyAsync(hidden_out, hcur, (size_t)Q * Hd * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (llama_out) {
        LinearCall pj{hcur, Q, Hd, {&w->llama_proj}, 1, c.proj_out};
        pj.y = llama_out;
        if (int rc = run_linear(h, pj, stream)) return rc;
    }
    return INFV_OK;
}

int infv_vqf_mean(const float* in, int32_t n, int64_t elems, float* out, void* stream) {
    if (!in || !out || n < 1 || elems < 1) return fail(INFV_ERR_INVALID, "bad argument");
    HIP_TRY(launch_qf_mean(in, n, elems, out, static_cast<hipStream_t>(stream)));
    return INFV_OK;
}

}  // extern "C"
