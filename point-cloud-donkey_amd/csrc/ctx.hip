// ctx.hip — context, scratch arena, device timers of libismhip.so
#include "common.h"
#include <cmath>
#include <cstring>
#include <cstdlib>

extern "C" void ism_cloud_pool_release(ismhip_ctx* ctx);

int ism_set_err(ismhip_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

void* ism_scratch(ismhip_ctx* ctx, int slot, size_t bytes) {
    if (bytes == 0) bytes = 16;
    auto& s = ctx->scratch[slot];
    if (s.bytes >= bytes) return s.p;
    if (s.p) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(s.p); s.p = nullptr; s.bytes = 0; }
    size_t want = bytes + bytes / 4;   // grow-only with headroom so steady state never reallocates
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) { ism_set_err(ctx, ISMHIP_ERR_NOMEM, "scratch hipMalloc failed"); return nullptr; }
    s.p = p; s.bytes = want;
    return p;
}

TimerScope::TimerScope(ismhip_ctx* c, const char* n) : ctx(c), name(n) {
    if (!ctx->timers_on) return;
    auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    a = get(); b = get();
    (void)hipEventRecord(a, ctx->stream);
}
TimerScope::~TimerScope() {
    if (!a) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->timers[name].pending.emplace_back(a, b);
}

extern "C" {

int ismhip_abi_version(void) { return ISMHIP_ABI_VERSION; }

static int ctx_create_impl(int device, void* stream, bool own, ismhip_ctx** out);

int ismhip_ctx_create(int device, void* stream, ismhip_ctx** out) { return ctx_create_impl(device, stream, stream == nullptr, out); }
int ismhip_ctx_create_on_stream(int device, void* stream, ismhip_ctx** out) { return ctx_create_impl(device, stream, false, out); }

static int ctx_create_impl(int device, void* stream, bool own, ismhip_ctx** out) {
    if (!out) return ISMHIP_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ISMHIP_ERR_NODEVICE;
    if (device < 0 || device >= n) return ISMHIP_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return ISMHIP_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ISMHIP_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ISMHIP_ERR_NODEVICE;   // code objects are gfx950 only
    ismhip_ctx* ctx = new ismhip_ctx();
    ctx->device = device;
    { const char* e = getenv("ISMHIP_KNN_MODE"); ctx->knn_mode = !e ? 0 : (!strcmp(e, "bf16x3") ? 1 : (!strcmp(e, "f32") ? 2 : 0)); }
    { const char* e = getenv("ISMHIP_XCD_MAP"); ctx->xcd_map = !(e && e[0] == '0'); }
    { const char* e = getenv("ISMHIP_GRID_XFRAC"); ctx->grid_xfrac = e ? (float)atof(e) : 0.f; }
    { const char* e = getenv("ISMHIP_SHOT_VAR"); ctx->shot_var = e ? atoi(e) : 0; }
    { const char* e = getenv("ISMHIP_KNN_TWOSTAGE"); ctx->knn_two_stage = !(e && e[0] == '0'); }
    { const char* e = getenv("ISMHIP_KNN_JOIN"); ctx->knn_join = !(e && e[0] == '0'); }
    { const char* e = getenv("ISMHIP_KNN_QPANEL"); ctx->knn_qpanel = e && e[0] == '1'; }
    { const char* e = getenv("ISMHIP_KNN_QPANEL2"); ctx->knn_qpanel2 = !(e && e[0] == '0'); }
    { const char* e = getenv("ISMHIP_KNN_HALF"); ctx->knn_half = e && e[0] == '1'; }
    { const char* e = getenv("ISMHIP_KNN_RING32"); ctx->knn_ring32 = e && e[0] == '1'; }
    { const char* e = getenv("ISMHIP_KNN_SPLITS"); ctx->knn_splits = e ? atoi(e) : 0; }
    { const char* e = getenv("ISMHIP_KNN_T"); ctx->knn_t = e ? atoi(e) : 0; }
    { const char* e = getenv("ISMHIP_KNN_DBG"); ctx->knn_dbg = e ? atoi(e) : 0; }
    { const char* e = getenv("ISMHIP_KNN_NORING"); ctx->knn_no_ring = e && e[0] == '1'; }
    { const char* e = getenv("ISMHIP_KNN_KB32"); ctx->knn_kb32 = e && e[0] == '1'; }
    { const char* e = getenv("ISMHIP_KNN_HELL_EMIT"); ctx->knn_hell_emit = !(e && e[0] == '0'); }
    { const char* e = getenv("ISMHIP_KNN_HELLINGER"); ctx->knn_hellinger = !(e && e[0] == '0'); }
    { const char* e = getenv("ISMHIP_KNN_T1"); if (e) ctx->knn_t1 = atoi(e); }
    { const char* e = getenv("ISMHIP_KNN_STAGE2_T4"); if (e) ctx->knn_stage2_t4 = atoi(e) != 0; }
    { const char* e = getenv("ISMHIP_KNN_PRE_STEP"); if (e && atoi(e) > 0) ctx->knn_pre_step = atoi(e); }
    { const char* e = getenv("ISMHIP_KNN_PRE_GAMMA"); if (e) ctx->knn_pre_gamma = (float)atof(e); }
    { const char* e = getenv("ISMHIP_KNN_PREPASS"); ctx->knn_prepass = !(e && e[0] == '0'); }
    { const char* e = getenv("ISMHIP_KNN_PCA_M"); ctx->knn_pca_m = e ? atoi(e) : -1; }
    { const char* e = getenv("ISMHIP_KNN_PCA_M2"); ctx->knn_pca_m2 = e ? atoi(e) : -1; }
    { const char* e = getenv("ISMHIP_KNN_TILE128"); ctx->knn_small_tile = e && e[0] == '1'; }
    if (!own) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return ISMHIP_ERR_HIP; }
        ctx->own_stream = true;
    }
    // RGB->CIELab look-up tables, reference: features/features_cshot.cpp:52-70
    std::vector<float> srgb(256), sxyz(4000);
    for (int i = 0; i < 256; i++) {
        float f = static_cast<float>(i) / 255.0f;
        srgb[i] = (f > 0.04045) ? powf((f + 0.055f) / 1.055f, 2.4f) : f / 12.92f;
    }
    for (int i = 0; i < 4000; i++) {
        float f = static_cast<float>(i) / 4000.0f;
        sxyz[i] = (f > 0.008856) ? static_cast<float>(powf(f, 0.3333f)) : static_cast<float>((7.787 * f) + (16.0 / 116.0));
    }
    if (hipMalloc((void**)&ctx->lut_srgb, 256 * 4) != hipSuccess || hipMalloc((void**)&ctx->lut_sxyz, 4000 * 4) != hipSuccess) {
        delete ctx; return ISMHIP_ERR_NOMEM;
    }
    (void)hipMemcpy(ctx->lut_srgb, srgb.data(), 256 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(ctx->lut_sxyz, sxyz.data(), 4000 * 4, hipMemcpyHostToDevice);
    if (hipMalloc((void**)&ctx->truncated_d, 4) != hipSuccess) { delete ctx; return ISMHIP_ERR_NOMEM; }
    (void)hipMemset(ctx->truncated_d, 0, 4);      // before the first launch on any stream: the device is idle
    (void)hipDeviceSynchronize();
    *out = ctx;
    return ISMHIP_OK;
}

int ismhip_ctx_destroy(ismhip_ctx* ctx) {
    if (!ctx) return ISMHIP_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ism_cloud_pool_release(ctx);
    for (auto& kv : ctx->scratch) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& kv : ctx->timers) for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->lut_srgb) (void)hipFree(ctx->lut_srgb);
    if (ctx->lut_sxyz) (void)hipFree(ctx->lut_sxyz);
    if (ctx->truncated_d) (void)hipFree(ctx->truncated_d);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return ISMHIP_OK;
}

int ismhip_sync(ismhip_ctx* ctx) {
    if (!ctx) return ISMHIP_ERR_INVALID;
    if (ctx->truncated_d) {                // caps of the maxima kernels (128 per object and class, 1024 per object): never silent
        // read and cleared ON the ctx stream: a null-stream memset is not ordered against an owned (non-blocking) stream and could wipe
        // the count of a find_maxima launched right after this call
        uint32_t n = 0;
        ISM_HIP(ctx, hipMemcpyAsync(&n, ctx->truncated_d, 4, hipMemcpyDeviceToHost, ctx->stream));
        ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (n) {
            ISM_HIP(ctx, hipMemsetAsync(ctx->truncated_d, 0, 4, ctx->stream));
            ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "find_maxima / hough3d_maxima: " + std::to_string(n) +
                               " (object, class) lists exceeded 128 maxima per class or 1024 per object since the last sync; the results of those objects are truncated");
        }
        return ISMHIP_OK;
    }
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ISMHIP_OK;
}

const char* ismhip_last_error(const ismhip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int ismhip_timers_enable(ismhip_ctx* ctx, int on) {
    if (!ctx) return ISMHIP_ERR_INVALID;
    ctx->timers_on = on != 0;
    return ISMHIP_OK;
}

static void resolve_timers(ismhip_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->timers) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { kv.second.ms += ms; kv.second.launches++; }
            ctx->event_pool.push_back(pr.first); ctx->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

int ismhip_timers_reset(ismhip_ctx* ctx) {
    if (!ctx) return ISMHIP_ERR_INVALID;
    resolve_timers(ctx);
    for (auto& kv : ctx->timers) { kv.second.ms = 0; kv.second.launches = 0; }
    return ISMHIP_OK;
}

int ismhip_timer_get(ismhip_ctx* ctx, const char* name, double* ms_out, int64_t* launches_out) {
    if (!ctx || !name) return ISMHIP_ERR_INVALID;
    resolve_timers(ctx);
    if (std::strcmp(name, "knn_pca_launches") == 0) { if (ms_out) *ms_out = (double)ctx->knn_pca_launches; if (launches_out) *launches_out = 1; return ISMHIP_OK; }
    if (std::strcmp(name, "knn_stage2_queries") == 0) { if (ms_out) *ms_out = (double)ctx->knn_stage2_queries; if (launches_out) *launches_out = 1; return ISMHIP_OK; }
    if (std::strcmp(name, "knn_flagged_queries") == 0 || std::strcmp(name, "knn_flagged_items") == 0) {     // counters, not times
        if (ms_out) *ms_out = (double)ctx->knn_stats[name[12] == 'q' ? 0 : 1];
        if (launches_out) *launches_out = 1;
        return ISMHIP_OK;
    }
    auto it = ctx->timers.find(name);
    if (ms_out) *ms_out = it == ctx->timers.end() ? 0.0 : it->second.ms;
    if (launches_out) *launches_out = it == ctx->timers.end() ? 0 : it->second.launches;
    return ISMHIP_OK;
}

}  // extern "C"
