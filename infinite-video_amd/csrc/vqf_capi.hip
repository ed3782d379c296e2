// C ABI of the video Q-former path (include/infv_vqf.h): workspace + launch sequence of one chunk.
#include "../../include/infv_vqf.h"
#include "capi_common.h"
#include <algorithm>
#include "vqf_internal.h"

#include <cstdlib>
#include <new>

using namespace infv;

struct infv_vqf_s {
    infv_vqf_config cfg;
    int dev = 0;
    bool exact_fp32 = false;
    // chunk-independent prefix of layer 0 (embedding LayerNorm, self-attention block, cross query, pre-multiplied query):
    // reused across encode_chunk calls while the host keeps the weights epoch unchanged (0 = never reuse)
    unsigned long long epoch = 0, c_epoch = 0;
    bool c_valid = false, c_qt_valid = false, c_qsplit_valid = false;
    DeviceBuf c_h1, c_xq, c_qt, c_qh, c_ql;
    // workspaces (grown on demand; a growing call synchronises the device first)
    DeviceBuf part, h_a, h_b, h1, h2, qkv, sa, xq, along, qt, S, O, merged, inter, kbar;
    DeviceBuf sFh, sFl, sTh, sTl, sPh, sPl, sQh, sQl;   // split-bf16 operands of the short-term attention
    DeviceBuf wFh, wFl, wTh, wTl;                       // the same split of a WHOLE video's frame tokens (layer-major path)
    bool fuse = true;                                   // one pass over the frame tokens: split + transpose + frame means
    double split_cache_gb = 64.0;                       // budget for a whole video's split tokens (INFV_VQF_SPLIT_CACHE_GB at create)
    // whole-video (layer-major) path
    DeviceBuf vA, v1, v2, vxq, valong, vshort, vmerged, vqkv, vsa, vinter, vu, vkbar, v_h1s, v_xqs;
    hipStream_t side = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    ~infv_vqf_s() {
        if (side) (void)hipStreamSynchronize(side);           // (a stream of the process-wide set: not destroyed here)
        if (ev_main) (void)hipEventDestroy(ev_main);
        if (ev_side) (void)hipEventDestroy(ev_side);
    }
};

namespace {

int check_cfg(const infv_vqf_config& c) {
    if (c.n_layers < 1 || c.n_layers > INFV_VQF_MAX_LAYERS) return fail(INFV_ERR_INVALID, "n_layers must be 1..%d", INFV_VQF_MAX_LAYERS);
    if (c.n_heads < 1 || c.hidden != c.n_heads * 64) return fail(INFV_ERR_UNSUPPORTED, "hidden must be n_heads * 64");
    // (the whole-layer entry points need n_query <= 32 and tokens_per_frame % 32 == 0 and check it themselves;
    //  infv_vqf_short_attention alone also serves the VideoChat2 shape: 96 query tokens, 196 tokens per frame)
    if (c.n_query < 1 || c.n_query > 256) return fail(INFV_ERR_UNSUPPORTED, "n_query must be 1..256");
    if (c.hidden % 64 || c.inter % 64 || c.enc_width % 32 || (c.proj_out % 64)) return fail(INFV_ERR_UNSUPPORTED, "widths must be multiples of 64 (enc_width: 32)");
    if (c.hidden > 4096 || c.inter > 4096 || c.proj_out > 4096 || 3 * c.hidden > 4096) return fail(INFV_ERR_UNSUPPORTED, "row widths above 4096 are not supported");
    if (c.nb_samples < 1) return fail(INFV_ERR_INVALID, "nb_samples must be >= 1");
    if (c.tokens_per_frame < 1) return fail(INFV_ERR_INVALID, "tokens_per_frame must be positive");
    return INFV_OK;
}

int check_layer_cfg(const infv_vqf_config& c) {       // what the fused per-layer kernels of encode_chunk / encode_video are built for
    if (c.n_query > 32) return fail(INFV_ERR_UNSUPPORTED, "encode_chunk / encode_video need n_query <= 32 (this handle: %d)", c.n_query);
    if (c.tokens_per_frame % 32) return fail(INFV_ERR_UNSUPPORTED, "encode_chunk / encode_video need tokens_per_frame %% 32 == 0");
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

// frames [nb][n_tokens][d]; xq: per-chunk [nb*Q][hidden] (shared_q = false) or one [Q][hidden] block used by every
// chunk (shared_q = true: layer 0 of the video Q-former, whose query does not depend on the chunk)
//   -> merged [nb*Q][hidden] = alpha * short-term context + (1 - alpha) * along   (along == nullptr: short-term only)
// Split-bf16 copies of frame tokens produced ahead of the attention (prepare_split): [.][n_tokens][d] and [.][d][n_tokens]
struct SplitRef { const __bf16 *Fh, *Fl, *Th, *Tl; };

static bool split_path(const infv_vqf_s* h, int n_tokens) {
    static const bool want_fp32 = [] { const char* e = getenv("INFV_VQF_FP32"); return e && atoi(e) != 0; }();
    return !want_fp32 && !h->exact_fp32 && h->cfg.enc_width % 64 == 0 && n_tokens % 64 == 0;   // (odd frame counts: exact-fp32 kernels)
}

// ONE pass over the frame tokens of `nb` chunks: hi/lo bf16 split, its transposed copy and (kbar != nullptr) the frame
// means the long-term memories consume.  The tokens do not depend on the layer, so every layer's short-term attention
// and every layer's memory share this pass (the reference reads them once per layer and once more for the pooling:
// Qformer.py:236, 278-291).  `whole` selects the video-sized buffers of the layer-major path.
static int prepare_split(infv_vqf_s* h, const float* frames, int nb, int n_tokens, float* kbar, bool whole, SplitRef* ref,
                         hipStream_t stream) {
    const int d = h->cfg.enc_width;
    const size_t szF = (size_t)nb * n_tokens * d * 2;
    DeviceBuf& Fh = whole ? h->wFh : h->sFh; DeviceBuf& Fl = whole ? h->wFl : h->sFl;
    DeviceBuf& Th = whole ? h->wTh : h->sTh; DeviceBuf& Tl = whole ? h->wTl : h->sTl;
    if (szF > Fh.bytes || szF > Fl.bytes || szF > Th.bytes || szF > Tl.bytes) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(Fh.reserve(szF)); HIP_TRY(Fl.reserve(szF)); HIP_TRY(Th.reserve(szF)); HIP_TRY(Tl.reserve(szF));
    }
    for (int c0 = 0; c0 < nb; c0 += 32768) {                   // grid.z limit
        const int n = nb - c0 < 32768 ? nb - c0 : 32768;
        const size_t o = (size_t)c0 * n_tokens * d;
        HIP_TRY(launch_split_transpose(frames + o, n, n_tokens, d, Fh.as<__bf16>() + o, Fl.as<__bf16>() + o, Th.as<__bf16>() + o,
                                       Tl.as<__bf16>() + o, stream,
                                       kbar ? kbar + (size_t)c0 * (n_tokens / h->cfg.tokens_per_frame) * d : nullptr,
                                       h->cfg.tokens_per_frame));
    }
    *ref = SplitRef{Fh.as<__bf16>(), Fl.as<__bf16>(), Th.as<__bf16>(), Tl.as<__bf16>()};
    return INFV_OK;
}

int short_attention(infv_vqf_s* h, const float* frames, int nb, int n_tokens, const float* xq, bool shared_q,
                    const infv_linear* key, const infv_linear* value, const float* along, float* merged,
                    hipStream_t stream, bool use_cache = false, const SplitRef* pre = nullptr,
                    hipEvent_t along_ready = nullptr /* `along` is produced on another stream: wait here, before the merge */) {
    const infv_vqf_config& c = h->cfg;
    const int Q = c.n_query, H = c.n_heads, d = c.enc_width, rows = H * Q;
    if (n_tokens < 32 || n_tokens % 32) return fail(INFV_ERR_INVALID, "n_tokens must be a positive multiple of 32");
    int kps = n_tokens;
    int sk = qf_pick_splitk_fill(rows, d, n_tokens, nb, &kps);
    if (split_path(h, n_tokens)) sk = split_gemm_pick_splitk(rows, d, n_tokens, nb, &kps);   // (64-deep k-tiles, its own tile shapes)
    const int nq = shared_q ? 1 : nb;
    // leading dimension of the score matrix: padded by 256 B so that its rows (the A operand of the second
    // contraction, one 128-B line per row per k-tile) do not all map to the same memory channel
    const long ldS = n_tokens + 64;
    const size_t needS = (size_t)nb * rows * ldS * sizeof(float);
    const size_t needO = (size_t)sk * nb * rows * d * sizeof(float);
    const size_t needQt = (size_t)nq * rows * d * sizeof(float);
    if (needS > h->S.bytes || needO > h->O.bytes || needQt > h->qt.bytes) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(h->S.reserve(needS)); HIP_TRY(h->O.reserve(needO)); HIP_TRY(h->qt.reserve(needQt));
    }
    float* qt = h->qt.as<float>();
    void* qh = h->sQh.p; void* ql = h->sQl.p;
    if (use_cache) {                                           // nq == 1: the cached query block of layer 0
        const size_t szq = (size_t)rows * d;
        if (szq * 4 > h->c_qt.bytes) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(h->c_qt.reserve(szq * 4)); HIP_TRY(h->c_qh.reserve(szq * 2)); HIP_TRY(h->c_ql.reserve(szq * 2));
            h->c_qt_valid = h->c_qsplit_valid = false;
        }
        qt = h->c_qt.as<float>(); qh = h->c_qh.p; ql = h->c_ql.p;
    }
    if (!use_cache || !h->c_qt_valid) {
        HIP_TRY(launch_qf_qtilde(xq, nq, Q, H, d, key->w, qt, stream));
        if (use_cache) { h->c_qt_valid = true; h->c_qsplit_valid = false; }
    }
    // Both big contractions ( [H*Q x d x n_tokens] each ) run as split-bf16 (three bf16 MFMA products, fp32 accumulate):
    // their rounding (~1e-5) only feeds the read-out.  INFV_VQF_FP32=1 selects the exact-fp32 MFMA kernels instead.
    if (split_path(h, n_tokens)) {
        const size_t szP = (size_t)nb * rows * n_tokens * 2, szQ = (size_t)nq * rows * d * 2;
        if (szP > h->sPh.bytes || szQ > h->sQh.bytes) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(h->sPh.reserve(szP)); HIP_TRY(h->sPl.reserve(szP)); HIP_TRY(h->sQh.reserve(szQ)); HIP_TRY(h->sQl.reserve(szQ));
        }
        SplitRef sr;
        if (pre) sr = *pre;                                   // the caller split these tokens already (shared by the layers)
        else if (int rc = prepare_split(h, frames, nb, n_tokens, nullptr, false, &sr, stream)) return rc;
        if (use_cache) {                                      // (the buffers may just have been (re)allocated above)
            qh = h->c_qh.p; ql = h->c_ql.p;
        } else {
            qh = h->sQh.p; ql = h->sQl.p;
        }
        if (!use_cache || !h->c_qsplit_valid) {
            HIP_TRY(launch_split_rows(qt, d, (long)nq * rows, d, qh, ql, d, stream));
            if (use_cache) h->c_qsplit_valid = true;
        }
        SplitGemm g{};                                        // S[b] = qt[b] . frames[b]^T
        g.A_hi = static_cast<const __bf16*>(qh); g.A_lo = static_cast<const __bf16*>(ql); g.lda = d; g.strideA = shared_q ? 0 : (long)rows * d;
        g.B_hi = sr.Fh; g.B_lo = sr.Fl; g.ldb = d; g.strideB = (long)n_tokens * d;
        g.C = h->S.as<float>(); g.ldc = ldS; g.strideC = (long)rows * ldS; g.split_stride = 0;
        g.M = rows; g.N = n_tokens; g.K = d; g.k_per_split = d; g.splitk = 1; g.nbatch = nb;
        HIP_TRY(launch_split_gemm(g, stream));
        HIP_TRY(launch_softmax_rows_split(h->S.as<float>(), (long)nb * rows, n_tokens, ldS, h->sPh.p, h->sPl.p, n_tokens, stream));
        SplitGemm p{};                                        // O[b] = P[b] . frames[b]
        p.A_hi = h->sPh.as<__bf16>(); p.A_lo = h->sPl.as<__bf16>(); p.lda = n_tokens; p.strideA = (long)rows * n_tokens;
        p.B_hi = sr.Th; p.B_lo = sr.Tl; p.ldb = n_tokens; p.strideB = (long)d * n_tokens;
        p.C = h->O.as<float>(); p.ldc = d; p.strideC = (long)rows * d; p.split_stride = (long)nb * rows * d;
        p.M = rows; p.N = d; p.K = n_tokens; p.k_per_split = kps; p.splitk = sk; p.nbatch = nb;
        HIP_TRY(launch_split_gemm(p, stream));
    } else {
    QfGemm g{};                                             // S[b] = qt[b] . frames[b]^T
    g.A = qt; g.lda = d; g.strideA = shared_q ? 0 : (long)rows * d;
    g.B[0] = frames; g.ldb = d; g.strideB = (long)n_tokens * d; g.seg_rows = n_tokens;
    g.C = h->S.as<float>(); g.ldc = ldS; g.strideC = (long)rows * ldS; g.split_stride = 0;
    g.M = rows; g.N = n_tokens; g.k_per_split = d; g.splitk = 1; g.nbatch = nb;
    HIP_TRY(launch_qf_gemm(g, false, stream));
    HIP_TRY(launch_qf_softmax_rows(h->S.as<float>(), (long)nb * rows, n_tokens, ldS, stream));
    QfGemm p{};                                             // O[b] = P[b] . frames[b]
    p.A = h->S.as<float>(); p.lda = ldS; p.strideA = (long)rows * ldS;
    p.B[0] = frames; p.ldb = d; p.strideB = (long)n_tokens * d; p.seg_rows = d;
    p.C = h->O.as<float>(); p.ldc = d; p.strideC = (long)rows * d; p.split_stride = (long)nb * rows * d;
    p.M = rows; p.N = d; p.k_per_split = kps; p.splitk = sk; p.nbatch = nb; p.K = n_tokens;
    HIP_TRY(launch_qf_gemm(p, true, stream));
    }
    const long o_split_stride = (long)nb * rows * d;
    // per-head value projection as a GEMM batched over (chunk, head): [Q x d] . Wv_h^T -> [Q x 64], then
    // bias + merge with the long-term context in the row epilogue (Qformer.py:298-304)
    HIP_TRY(launch_qf_sum_slabs(h->O.as<float>(), sk, o_split_stride, (long)nb * rows * d, stream));
    const int M2 = nb * Q, hidden = c.hidden;
    int sk2 = 8;
    while (sk2 > 1 && (d % (32 * sk2) || nb * H * sk2 > 4096)) sk2 >>= 1;
    const size_t needP = (size_t)sk2 * M2 * hidden * sizeof(float);
    if (needP > h->part.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->part.reserve(needP)); }
    QfGemm v{};
    v.A = h->O.as<float>(); v.lda = d; v.strideA = (long)Q * d; v.strideA2 = (long)rows * d;   // inner = head, outer = chunk
    v.B[0] = value->w; v.ldb = d; v.strideB = 64L * d; v.strideB2 = 0; v.seg_rows = 64;
    v.C = h->part.as<float>(); v.ldc = hidden; v.strideC = 64; v.strideC2 = (long)Q * hidden;
    v.split_stride = (long)M2 * hidden;
    v.M = Q; v.N = 64; v.k_per_split = d / sk2; v.splitk = sk2; v.nbatch = nb * H; v.inner = H;
    HIP_TRY(launch_qf_gemm(v, false, stream));
    QfEpilogue e{};
    e.parts = h->part.as<float>(); e.nsplit = sk2; e.split_stride = (long)M2 * hidden; e.ld_in = hidden;
    e.bias[0] = value->b; e.seg_cols = hidden; e.act = QF_ACT_NONE;
    e.scale = along ? c.alpha : 1.f; e.res_scale = (float)(1.0 - (double)c.alpha);
    e.residual = along; e.ld_res = hidden; e.res_rows = M2;
    e.out = merged; e.ld_out = hidden; e.M = M2; e.width = hidden; e.eps = c.ln_eps;
    if (along_ready) HIP_TRY(hipStreamWaitEvent(stream, along_ready, 0));
    HIP_TRY(launch_qf_epilogue(e, stream));
    return INFV_OK;
}

int ensure_streams(infv_vqf_s* h) {
    if (h->side) return INFV_OK;
    // one of the LTM library's shared worker streams rather than a stream of its own: the runtime multiplexes a process's
    // streams onto 4 hardware queues, and a fifth stream that lands on the caller's queue serialises the two schedules
    // (layer-major path inside bench.py: 0.26 instead of 0.155 ms per chunk)
    if (int rc = shared_worker_stream(&h->side)) return rc;
    HIP_TRY(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
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
    {   // INFV_VQF_FUSE=0 (read per handle, for A/B tests): separate pooling pass + one split pass per layer, as in round 1
        const char* e = getenv("INFV_VQF_FUSE");
        h->fuse = !e || atoi(e) != 0;
        if (const char* g = getenv("INFV_VQF_SPLIT_CACHE_GB")) h->split_cache_gb = atof(g);
    }
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

int infv_vqf_set_precision(infv_vqf_handle h, int32_t exact_fp32) {
    if (!h) return fail(INFV_ERR_INVALID, "null handle");
    h->exact_fp32 = exact_fp32 != 0;
    return INFV_OK;
}

int infv_vqf_set_weights_epoch(infv_vqf_handle h, uint64_t epoch) {
    if (!h) return fail(INFV_ERR_INVALID, "null handle");
    h->epoch = epoch;
    return INFV_OK;
}

int infv_vqf_short_attention(infv_vqf_handle h, const float* frames, int32_t n_tokens, const float* xq,
                             const infv_linear* key, const infv_linear* value, const float* a_long,
                             float* merged, void* stream) {
    if (!h || !frames || !xq || !key || !value || !merged || !key->w || !value->w || !value->b)
        return fail(INFV_ERR_INVALID, "null argument");
    return short_attention(h, frames, 1, n_tokens, xq, false, key, value, a_long, merged, static_cast<hipStream_t>(stream));
}

int infv_vqf_encode_chunk(infv_vqf_handle h, const infv_ltm_handle* ltm, const float* frames, int32_t T,
                          const infv_vqf_weights* w, const double* u, int32_t new_video,
                          float* hidden_out, float* llama_out, void* stream_) {
    if (!h || !frames || !w) return fail(INFV_ERR_INVALID, "null argument");
    const infv_vqf_config& c = h->cfg;
    if (int rc = check_layer_cfg(c)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool use_ltm = c.alpha != 1.0f;                      // Qformer.py:220-223
    if (use_ltm && !ltm) return fail(INFV_ERR_INVALID, "alpha != 1 needs the per-layer LTM handles");
    if (T < 1) return fail(INFV_ERR_INVALID, "T must be >= 1");
    if (llama_out && (c.proj_out <= 0 || !w->llama_proj.w)) return fail(INFV_ERR_INVALID, "llama_out without llama_proj");
    const int Q = c.n_query, Hd = c.hidden, n_tokens = T * c.tokens_per_frame;
    if (use_ltm)
        if (int rc = ensure_streams(h)) return rc;

    // The prefix of layer 0 (embedding LayerNorm -> self-attention block -> cross query) depends on the weights only,
    // not on the chunk: with a non-zero weights epoch it is computed once and reused until the epoch changes.
    const bool caching = h->epoch != 0;
    const bool prefix_cached = caching && h->c_valid && h->c_epoch == h->epoch;
    if (caching && !prefix_cached) {
        const size_t sz = (size_t)Q * Hd * sizeof(float);
        if (sz > h->c_h1.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->c_h1.reserve(sz)); HIP_TRY(h->c_xq.reserve(sz)); }
        h->c_valid = false; h->c_qt_valid = false; h->c_qsplit_valid = false;
    }
    if (!prefix_cached) {
        // embeddings: LayerNorm of the learned query tokens (Qformer.py:108-112)
        QfEpilogue e{};
        e.parts = w->query_tokens; e.nsplit = 1; e.split_stride = 0; e.ld_in = Hd; e.seg_cols = Hd;
        e.gamma = w->emb_ln.gamma; e.beta = w->emb_ln.beta; e.eps = c.ln_eps; e.scale = 1.f; e.res_scale = 1.f;
        e.out = h->h_a.as<float>(); e.ld_out = Hd; e.M = Q; e.width = Hd; e.res_rows = 1;
        HIP_TRY(launch_qf_epilogue(e, stream));
    }
    float* hcur = h->h_a.as<float>();
    float* hnext = h->h_b.as<float>();

    if (use_ltm) {
        const size_t need = (size_t)T * c.enc_width * sizeof(float);
        if (need > h->kbar.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->kbar.reserve(need)); }
        if (new_video)
            for (int l = 0; l < c.n_layers; ++l)
                if (int rc = infv_ltm_reset(ltm[l])) return rc;
    }
    // one pass over the chunk's tokens serves the pooling and the split operands of every layer's short-term attention
    SplitRef sref; const SplitRef* pre = nullptr;
    const bool fused_pool = h->fuse && split_path(h, n_tokens) && 64 % c.tokens_per_frame == 0;
    if (h->fuse && split_path(h, n_tokens)) {
        if (int rc = prepare_split(h, frames, 1, n_tokens, use_ltm && fused_pool ? h->kbar.as<float>() : nullptr, false, &sref, stream)) return rc;
        pre = &sref;
    }
    if (use_ltm && !fused_pool) {
        // (the frames of this entry point are fp32: the handle's token dtype is sticky state an earlier bf16 caller may have set)
        if (int rc = infv_ltm_set_token_dtype(ltm[0], INFV_TOKENS_F32)) return rc;
        if (int rc = infv_ltm_pool(ltm[0], frames, T, h->kbar.as<float>(), stream_)) return rc;   // shared by all layers
    }

    for (int l = 0; l < c.n_layers; ++l) {
        const infv_vqf_layer& L = w->layer[l];
        const bool l0c = caching && l == 0;                   // this layer's prefix lives in the cache buffers
        float* h1 = l0c ? h->c_h1.as<float>() : h->h1.as<float>();
        float* xqb = l0c ? h->c_xq.as<float>() : h->xq.as<float>();
        if (!(l0c && prefix_cached)) {
            // ---- self-attention over the query tokens + output (Qformer.py:442-470 -> BertAttention)
            LinearCall qkv{hcur, Q, Hd, {&L.self_q, &L.self_k, &L.self_v}, 3, Hd};
            qkv.y = h->qkv.as<float>();
            if (int rc = run_linear(h, qkv, stream)) return rc;
            HIP_TRY(launch_qf_self_attention(h->qkv.as<float>(), 1, Q, c.n_heads, h->sa.as<float>(), stream));
            LinearCall so{h->sa.as<float>(), Q, Hd, {&L.self_o}, 1, Hd};
            so.residual = hcur; so.res_rows = Q; so.ln = &L.self_ln; so.y = h1;
            if (int rc = run_linear(h, so, stream)) return rc;
            // ---- cross-attention: query, long-term memory, short-term attention, merge, output
            LinearCall xq{h1, Q, Hd, {&L.x_q}, 1, Hd};
            xq.y = xqb;
            if (int rc = run_linear(h, xq, stream)) return rc;
            if (l0c) { h->c_valid = true; h->c_epoch = h->epoch; }
        }
        const float* along = nullptr;
        hipEvent_t along_ready = nullptr;
        if (use_ltm) {
            // The memory step and the short-term attention both start from the cross query and meet only in the merge
            // (Qformer.py:216-223 vs :225-302, :303-304): the step's three launches go to the shared worker stream and run
            // beside the attention's contractions; the merge epilogue waits for them.
            infv_ltm_proj pr{};
            pr.wk = L.x_k.w; pr.bk = L.x_k.b; pr.wv = L.x_v.w; pr.bv = L.x_v.b;
            const double* ul = u ? u + (size_t)l * c.nb_samples : nullptr;
            HIP_TRY(hipEventRecord(h->ev_main, stream));              // cross query, pooled frames, the previous layer's merge
            HIP_TRY(hipStreamWaitEvent(h->side, h->ev_main, 0));
            if (int rc = infv_ltm_step(ltm[l], h->kbar.as<float>(), T, xqb, Q, &pr, ul,
                                       h->along.as<float>(), h->side)) return rc;
            HIP_TRY(hipEventRecord(h->ev_side, h->side));
            along = h->along.as<float>();
            along_ready = h->ev_side;
        }
        if (int rc = short_attention(h, frames, 1, n_tokens, xqb, false, &L.x_k, &L.x_v, along,
                                     h->merged.as<float>(), stream, l0c, pre, along_ready)) return rc;
        LinearCall xo{h->merged.as<float>(), Q, Hd, {&L.x_o}, 1, Hd};
        xo.residual = h1; xo.res_rows = Q; xo.ln = &L.x_ln; xo.y = h->h2.as<float>();
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

int infv_vqf_encode_video(infv_vqf_handle h, const infv_ltm_handle* ltm, const float* frames, int32_t n_chunks,
                          int32_t T, const infv_vqf_weights* w, const double* u, int32_t new_video,
                          float* hidden_out, float* llama_out, float* llama_mean, void* stream_) {
    if (!h || !frames || !w || n_chunks < 1) return fail(INFV_ERR_INVALID, "bad argument");
    const infv_vqf_config& c = h->cfg;
    if (int rc = check_layer_cfg(c)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool use_ltm = c.alpha != 1.0f;
    if (use_ltm && !ltm) return fail(INFV_ERR_INVALID, "alpha != 1 needs the per-layer LTM handles");
    if (T < 1) return fail(INFV_ERR_INVALID, "T must be >= 1");
    if ((llama_out || llama_mean) && (c.proj_out <= 0 || !w->llama_proj.w)) return fail(INFV_ERR_INVALID, "llama output without llama_proj");
    if (int rc = ensure_streams(h)) return rc;
    const int Q = c.n_query, Hd = c.hidden, C = n_chunks, S = c.nb_samples;
    const int n_tokens = T * c.tokens_per_frame;
    const long chunk_k = (long)n_tokens * c.enc_width;
    const long M = (long)C * Q;                               // rows of the whole-video activations
    // ---- workspaces: whole-video activations + per-block scratch ----
    const int RB = C < 64 ? C : 64;                           // chunks per row block of the query-token GEMMs
    // chunks per sub-batch of the short-term attention: the second contraction has few, long tiles per chunk
    // (rows/128 x d/128 = 18 at the headline shape, K = T*P), so pick the count whose tile total fills whole rounds
    // of the 256 CUs (16 chunks = 288 tiles ran at 56 % of 14 chunks' rate per tile)
    int NB = C;
    if (C > 8) {
        auto fill = [](long wgs) { return (double)wgs / (double)(((wgs + 255) / 256) * 256); };
        // with the 384 x 256 kernel (split path, whole tiles): workgroups of the scores contraction, and of the read-out with the
        // split-K count that fills best -- both should come out as whole rounds (headline: 32 chunks = 1024 and 96 x 8)
        const long ts = split_path(h, n_tokens) ? split_gemm_wide_tile_count(c.n_heads * Q, n_tokens) : 0;
        const long tr = split_path(h, n_tokens) ? split_gemm_wide_tile_count(c.n_heads * Q, c.enc_width) : 0;
        const int tiles = ((c.n_heads * Q + 127) / 128) * ((c.enc_width + 127) / 128);
        double best = -1.0;
        for (int nb = 8; nb <= 32 && nb <= C; ++nb) {
            double eff;
            if (ts > 0 && tr > 0) {
                double er = 0.0;
                for (int sk = 1; sk <= 16; ++sk) er = std::max(er, fill(tr * nb * sk) - 0.004 * sk);
                eff = 0.5 * (fill(ts * nb) + er);
            } else {
                eff = fill((long)tiles * nb);
            }
            if (eff >= best - 1e-9) { best = eff; NB = nb; }
        }
    }
    {
        static const int nb_env = [] { const char* e = exp_env("INFV_VQF_NB"); return e ? atoi(e) : 0; }();   // (sweeps)
        if (nb_env > 0) NB = nb_env < C ? nb_env : C;
    }
    {
        const size_t act = (size_t)M * Hd * sizeof(float);
        bool grow = act > h->vA.bytes || (size_t)RB * Q * 3 * Hd * sizeof(float) > h->vqkv.bytes ||
                    (size_t)RB * Q * c.inter * sizeof(float) > h->vinter.bytes ||
                    (use_ltm && ((size_t)C * S * sizeof(double) > h->vu.bytes || (size_t)C * T * c.enc_width * sizeof(float) > h->vkbar.bytes));
        if (grow) HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(h->vA.reserve(act)); HIP_TRY(h->v1.reserve(act)); HIP_TRY(h->v2.reserve(act)); HIP_TRY(h->vxq.reserve(act));
        HIP_TRY(h->vshort.reserve(act)); HIP_TRY(h->vmerged.reserve(act));
        if (use_ltm) {
            HIP_TRY(h->valong.reserve(act));
            HIP_TRY(h->vu.reserve((size_t)C * S * sizeof(double)));
            HIP_TRY(h->vkbar.reserve((size_t)C * T * c.enc_width * sizeof(float)));
        }
        HIP_TRY(h->vqkv.reserve((size_t)RB * Q * 3 * Hd * sizeof(float)));
        HIP_TRY(h->vsa.reserve((size_t)RB * Q * Hd * sizeof(float)));
        HIP_TRY(h->vinter.reserve((size_t)RB * Q * c.inter * sizeof(float)));
        HIP_TRY(h->v_h1s.reserve((size_t)Q * Hd * sizeof(float)));
        HIP_TRY(h->v_xqs.reserve((size_t)Q * Hd * sizeof(float)));
    }
    // ---- ONE pass over the whole video's frame tokens: split-bf16 operands of every layer's short-term attention and
    //      the frame means of every layer's memory.  The copies take 2 x the tokens' bytes (12.7 GB for the headline's
    //      252 chunks): kept for the call when they fit INFV_VQF_SPLIT_CACHE_GB (default 64 of the 288 GB). ----
    SplitRef wref{}; bool have_w = false, have_kbar = false;
    {
        const double budget_gb = h->split_cache_gb;
        const double need_gb = 4.0 * (double)C * (double)chunk_k * 2.0 / 1e9;
        if (h->fuse && split_path(h, n_tokens) && need_gb <= budget_gb) {
            have_kbar = use_ltm && 64 % c.tokens_per_frame == 0;
            if (int rc = prepare_split(h, frames, C, n_tokens, have_kbar ? h->vkbar.as<float>() : nullptr, true, &wref, stream)) return rc;
            have_w = true;
        }
    }
    auto pre_at = [&](int c0, SplitRef* r) -> const SplitRef* {  // the cached split of chunks c0.. (nullptr: split per sub-batch)
        if (!have_w) return nullptr;
        const long o = (long)c0 * chunk_k;
        *r = SplitRef{wref.Fh + o, wref.Fl + o, wref.Th + o, wref.Tl + o};
        return r;
    };
    float* vA = h->vA.as<float>(); float* v1 = h->v1.as<float>(); float* v2 = h->v2.as<float>();
    float* vxq = h->vxq.as<float>(); float* valong = h->valong.as<float>();
    float* vshort = h->vshort.as<float>(); float* vmerged = h->vmerged.as<float>();

    // embeddings: LayerNorm of the learned query tokens (chunk-independent)
    QfEpilogue e0{};
    e0.parts = w->query_tokens; e0.nsplit = 1; e0.ld_in = Hd; e0.seg_cols = Hd; e0.scale = 1.f; e0.res_scale = 1.f;
    e0.gamma = w->emb_ln.gamma; e0.beta = w->emb_ln.beta; e0.eps = c.ln_eps;
    e0.out = h->h_a.as<float>(); e0.ld_out = Hd; e0.M = Q; e0.width = Hd; e0.res_rows = 1;
    HIP_TRY(launch_qf_epilogue(e0, stream));

    auto ltm_u = [&](int l) -> int {                          // u[:, l, :] -> contiguous [C][S] for a one-layer handle
        if (!u) return INFV_OK;
        HIP_TRY(hipMemcpy2DAsync(h->vu.p, (size_t)S * sizeof(double), u + (size_t)l * S, (size_t)c.n_layers * S * sizeof(double),
                                 (size_t)S * sizeof(double), (size_t)C, hipMemcpyDeviceToDevice, stream));
        return INFV_OK;
    };

    for (int l = 0; l < c.n_layers; ++l) {
        const infv_vqf_layer& L = w->layer[l];
        infv_ltm_proj pr{};
        pr.wk = L.x_k.w; pr.bk = L.x_k.b; pr.wv = L.x_v.w; pr.bv = L.x_v.b;
        const bool shared = (l == 0);                         // layer 0: the hidden states entering it do not depend on the chunk
        const float* res1;                                    // residual / input of the cross-attention block
        int res1_rows;
        if (shared) {
            // ---- self-attention block + cross query once ----
            LinearCall qkv{h->h_a.as<float>(), Q, Hd, {&L.self_q, &L.self_k, &L.self_v}, 3, Hd};
            qkv.y = h->qkv.as<float>();
            if (int rc = run_linear(h, qkv, stream)) return rc;
            HIP_TRY(launch_qf_self_attention(h->qkv.as<float>(), 1, Q, c.n_heads, h->sa.as<float>(), stream));
            LinearCall so{h->sa.as<float>(), Q, Hd, {&L.self_o}, 1, Hd};
            so.residual = h->h_a.as<float>(); so.res_rows = Q; so.ln = &L.self_ln; so.y = h->v_h1s.as<float>();
            if (int rc = run_linear(h, so, stream)) return rc;
            LinearCall xq{h->v_h1s.as<float>(), Q, Hd, {&L.x_q}, 1, Hd};
            xq.y = h->v_xqs.as<float>();
            if (int rc = run_linear(h, xq, stream)) return rc;
            res1 = h->v_h1s.as<float>(); res1_rows = Q;
            // ---- long-term memory of every chunk with the constant query: the whole-video fast path ----
            if (use_ltm) {
                if (int rc = ltm_u(l)) return rc;
                if (have_kbar) {
                    if (int rc = infv_ltm_consolidate_pooled(ltm[l], h->vkbar.as<float>(), C, T, h->v_xqs.as<float>(), Q, &pr,
                                                             u ? h->vu.as<double>() : nullptr, new_video, valong, stream_)) return rc;
                } else if (int rc = infv_ltm_set_token_dtype(ltm[l], INFV_TOKENS_F32)) {
                    return rc;
                } else if (int rc = infv_ltm_consolidate(ltm[l], frames, C, T, h->v_xqs.as<float>(), Q, &pr,
                                                         u ? h->vu.as<double>() : nullptr, new_video, valong, stream_)) return rc;
            }
            // ---- short-term attention, merged with the long-term context ----
            for (int c0 = 0; c0 < C; c0 += NB) {
                const int nb = C - c0 < NB ? C - c0 : NB;
                SplitRef sr;
                if (int rc = short_attention(h, frames + c0 * chunk_k, nb, n_tokens, h->v_xqs.as<float>(), true, &L.x_k, &L.x_v,
                                             use_ltm ? valong + (long)c0 * Q * Hd : nullptr, vmerged + (long)c0 * Q * Hd, stream,
                                             false, pre_at(c0, &sr))) return rc;
            }
        } else {
            // ---- self-attention block + cross query of every chunk (row blocks) ----
            for (int c0 = 0; c0 < C; c0 += RB) {
                const int nb = C - c0 < RB ? C - c0 : RB;
                const long r0 = (long)c0 * Q * Hd;
                LinearCall qkv{vA + r0, nb * Q, Hd, {&L.self_q, &L.self_k, &L.self_v}, 3, Hd};
                qkv.y = h->vqkv.as<float>();
                if (int rc = run_linear(h, qkv, stream)) return rc;
                HIP_TRY(launch_qf_self_attention(h->vqkv.as<float>(), nb, Q, c.n_heads, h->vsa.as<float>(), stream));
                LinearCall so{h->vsa.as<float>(), nb * Q, Hd, {&L.self_o}, 1, Hd};
                so.residual = vA + r0; so.res_rows = nb * Q; so.ln = &L.self_ln; so.y = v1 + r0;
                if (int rc = run_linear(h, so, stream)) return rc;
                LinearCall xq{v1 + r0, nb * Q, Hd, {&L.x_q}, 1, Hd};
                xq.y = vxq + r0;
                if (int rc = run_linear(h, xq, stream)) return rc;
            }
            res1 = v1; res1_rows = (int)M;
            // ---- long-term memory with per-chunk queries: the sequential per-call chain, on the side stream, while
            //      the main stream runs this layer's short-term attention ----
            if (use_ltm) {
                if (int rc = ltm_u(l)) return rc;
                HIP_TRY(hipEventRecord(h->ev_main, stream));
                HIP_TRY(hipStreamWaitEvent(h->side, h->ev_main, 0));
                if (new_video)
                    if (int rc = infv_ltm_reset(ltm[l])) return rc;
                if (!have_kbar) {
                    if (int rc = infv_ltm_set_token_dtype(ltm[l], INFV_TOKENS_F32)) return rc;
                    if (int rc = infv_ltm_pool(ltm[l], frames, (int64_t)C * T, h->vkbar.as<float>(), h->side)) return rc;
                }
                // per-chunk queries: new-row projections of all chunks in one GEMM, then the chain chunk by chunk
                if (int rc = infv_ltm_steps(ltm[l], h->vkbar.as<float>(), C, T, vxq, Q, &pr, u ? h->vu.as<double>() : nullptr,
                                            valong, h->side)) return rc;
                HIP_TRY(hipEventRecord(h->ev_side, h->side));
            }
            for (int c0 = 0; c0 < C; c0 += NB) {
                const int nb = C - c0 < NB ? C - c0 : NB;
                SplitRef sr;
                if (int rc = short_attention(h, frames + c0 * chunk_k, nb, n_tokens, vxq + (long)c0 * Q * Hd, false, &L.x_k, &L.x_v,
                                             nullptr, (use_ltm ? vshort : vmerged) + (long)c0 * Q * Hd, stream, false,
                                             pre_at(c0, &sr))) return rc;
            }
            if (use_ltm) {
                HIP_TRY(hipStreamWaitEvent(stream, h->ev_side, 0));
                for (long m0 = 0; m0 < M; m0 += 32768) {       // merged = alpha * short + (1 - alpha) * long
                    QfEpilogue em{};
                    em.parts = vshort + m0 * Hd; em.nsplit = 1; em.ld_in = Hd; em.seg_cols = Hd;
                    em.scale = c.alpha; em.res_scale = (float)(1.0 - (double)c.alpha);
                    em.residual = valong + m0 * Hd; em.ld_res = Hd; em.res_rows = (int)(M - m0 < 32768 ? M - m0 : 32768);
                    em.out = vmerged + m0 * Hd; em.ld_out = Hd; em.M = em.res_rows; em.width = Hd; em.eps = c.ln_eps;
                    HIP_TRY(launch_qf_epilogue(em, stream));
                }
            }
        }
        // ---- cross-attention output + query FFN of every chunk (row blocks) ----
        for (int c0 = 0; c0 < C; c0 += RB) {
            const int nb = C - c0 < RB ? C - c0 : RB;
            const long r0 = (long)c0 * Q * Hd;
            LinearCall xo{vmerged + r0, nb * Q, Hd, {&L.x_o}, 1, Hd};
            xo.residual = shared ? res1 : res1 + r0; xo.res_rows = shared ? res1_rows : nb * Q; xo.ln = &L.x_ln; xo.y = v2 + r0;
            if (int rc = run_linear(h, xo, stream)) return rc;
            LinearCall fi{v2 + r0, nb * Q, Hd, {&L.ffn_in}, 1, c.inter};
            fi.act = QF_ACT_GELU; fi.y = h->vinter.as<float>();
            if (int rc = run_linear(h, fi, stream)) return rc;
            LinearCall fo{h->vinter.as<float>(), nb * Q, c.inter, {&L.ffn_out}, 1, Hd};
            fo.residual = v2 + r0; fo.res_rows = nb * Q; fo.ln = &L.ffn_ln; fo.y = vA + r0;
            if (int rc = run_linear(h, fo, stream)) return rc;
        }
    }
    if (hidden_out)
        HIP_TRY(hipMemcpyAsync(hidden_out, vA, (size_t)M * Hd * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (llama_out || llama_mean) {
        float* lo = llama_out;
        if (!lo) {                                             // only the mean is wanted: project into scratch
            const size_t need = (size_t)M * c.proj_out * sizeof(float);
            if (need > h->S.bytes) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(h->S.reserve(need)); }
            lo = h->S.as<float>();
        }
        for (int c0 = 0; c0 < C; c0 += RB) {
            const int nb = C - c0 < RB ? C - c0 : RB;
            LinearCall pj{vA + (long)c0 * Q * Hd, nb * Q, Hd, {&w->llama_proj}, 1, c.proj_out};
            pj.y = lo + (long)c0 * Q * c.proj_out;
            if (int rc = run_linear(h, pj, stream)) return rc;
        }
        if (llama_mean) HIP_TRY(launch_qf_mean(lo, C, (long)Q * c.proj_out, llama_mean, stream));
    }
    return INFV_OK;
}

int infv_vqf_mean(const float* in, int32_t n, int64_t elems, float* out, void* stream) {
    if (!in || !out || n < 1 || elems < 1) return fail(INFV_ERR_INVALID, "bad argument");
    HIP_TRY(launch_qf_mean(in, n, elems, out, static_cast<hipStream_t>(stream)));
    return INFV_OK;
}

}  // extern "C"
