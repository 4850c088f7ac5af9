// ism3d.cpp — bodies of the C++ host mirror (ism3d.h). Every hot-path body is a call into libismhip.so (C ABI);
// this file only owns configuration, batching, device buffers and the training-side bookkeeping the reference does on
// the host (reference files cited per function, relative to /root/reference/src/implicit_shape_model).
#include "ism3d.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <numeric>
#include <random>

namespace ism3d {

// ---------------------------------------------------------------------------------------------------------------
// logging (reference: log4cxx macros utils/utils.h:27-38; INFO<->WARN switch implicit_shape_model.cpp:145-151)
// ---------------------------------------------------------------------------------------------------------------
static bool g_log_info = true;
#define LOG_INFO(x) do { if (g_log_info) std::cout << "INFO: " << x << std::endl; } while (0)
#define LOG_WARN(x) do { std::cout << "WARN: " << x << std::endl; } while (0)
#define LOG_ERROR(x) do { std::cerr << "ERROR: " << x << std::endl; } while (0)
void jsonWarnMissing(const std::string& name) { LOG_WARN("parameter \"" << name << "\" not found, using default"); }

// ---------------------------------------------------------------------------------------------------------------
// device plumbing
// ---------------------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reserve(size_t n) {
        if (n <= bytes) return;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        size_t want = n + n / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) throw RuntimeException("hipMalloc failed");
        bytes = want;
    }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(bytes, o.bytes); }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
    template <typename T> const T* as() const { return reinterpret_cast<const T*>(p); }
};

struct DeviceFeatures {
    std::vector<uint32_t> off{0};     // per-object ranges of the kept features
    int dim = 0;
    uint32_t n = 0;
    DevBuf desc, lrf, kx, ky, kz, src;
};

class DeviceSession {
public:
    ismhip_ctx* ctx = nullptr;
    int n_obj = 0;
    std::vector<uint32_t> pt_off, kp_off;
    DevBuf x, y, z, nx, ny, nz, rgba, kx, ky, kz, krgba;
    DevBuf fx, fy, fz, fnx, fny, fnz, frgba;      // second set of point arrays: target of ismhip_filter_normals, swapped in afterwards
    ismhip_cloud* cloud = nullptr;
    bool has_color = false;
    // vote space of the current batch (Voting::m_votes)
    DevBuf v_pos, v_w, v_cls, v_inst, v_cw, v_bs, v_bq, idx, dist;
    DevBuf obj_cen, obj_rad;              // per-object cloud centroid / farthest point (single-object max types)
    std::vector<uint32_t> slot_off;
    size_t n_slots = 0;
    int n_classes = 0;
    // scratch for raw (uncompacted) features
    DevBuf raw_lrf, raw_desc, raw_cnt;

    explicit DeviceSession(int device) {
        int rc = ismhip_ctx_create(device, nullptr, &ctx);
        if (rc == ISMHIP_ERR_NODEVICE) throw RuntimeException("no gfx950 device available: the recognition path has no CPU fallback");
        if (rc != ISMHIP_OK) throw RuntimeException("ismhip_ctx_create failed");
    }
    ~DeviceSession() {
        if (cloud) ismhip_cloud_destroy(ctx, cloud);
        if (ctx) ismhip_ctx_destroy(ctx);
    }
    void check(int rc, const char* what) const {
        if (rc != ISMHIP_OK) throw RuntimeException(std::string(what) + " failed (" + std::to_string(rc) + "): " + ismhip_last_error(ctx));
    }
    void sync() const { check(ismhip_sync(ctx), "ismhip_sync"); }
    template <typename T> static void h2d(DevBuf& b, const std::vector<T>& v) {
        b.reserve(std::max<size_t>(v.size(), 1) * sizeof(T));
        if (!v.empty() && hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw RuntimeException("hipMemcpy H2D failed");
    }
    template <typename T> void d2h(std::vector<T>& v, const DevBuf& b, size_t n) const {
        sync();
        v.resize(n);
        if (n && hipMemcpy(v.data(), b.p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) throw RuntimeException("hipMemcpy D2H failed");
    }
    // concatenates the objects into SoA arrays, uploads them and builds the search surface
    // kps == nullptr: voxel-grid keypoints are computed on the device (ismhip_voxel_keypoints) with leaf `device_leaf`
    void uploadBatch(const std::vector<const PointCloud*>& clouds, const std::vector<KeypointSet>* kps_p, float cell, bool with_color, float device_leaf = 0.f) {
        static const std::vector<KeypointSet> no_kps;
        const bool dev_kp = kps_p == nullptr;
        n_obj = (int)clouds.size();
        const std::vector<KeypointSet>& kps = dev_kp ? no_kps : *kps_p;
        static const KeypointSet empty_set;
        pt_off.assign(1, 0); kp_off.assign(1, 0);
        std::vector<float> hx, hy, hz, hnx, hny, hnz, hkx, hky, hkz;
        std::vector<uint32_t> hrgba, hkrgba;
        for (int o = 0; o < n_obj; ++o) {
            const PointCloud& c = *clouds[o];
            hx.insert(hx.end(), c.x.begin(), c.x.end()); hy.insert(hy.end(), c.y.begin(), c.y.end()); hz.insert(hz.end(), c.z.begin(), c.z.end());
            hnx.insert(hnx.end(), c.nx.begin(), c.nx.end()); hny.insert(hny.end(), c.ny.begin(), c.ny.end()); hnz.insert(hnz.end(), c.nz.begin(), c.nz.end());
            if (with_color) {
                if (c.rgba.size() == c.size()) hrgba.insert(hrgba.end(), c.rgba.begin(), c.rgba.end());
                else hrgba.insert(hrgba.end(), c.size(), 0u);
            }
            pt_off.push_back((uint32_t)hx.size());
            const KeypointSet& k = dev_kp ? empty_set : kps[o];
            hkx.insert(hkx.end(), k.x.begin(), k.x.end()); hky.insert(hky.end(), k.y.begin(), k.y.end()); hkz.insert(hkz.end(), k.z.begin(), k.z.end());
            if (with_color) {
                if (k.rgba.size() == k.size()) hkrgba.insert(hkrgba.end(), k.rgba.begin(), k.rgba.end());
                else hkrgba.insert(hkrgba.end(), k.size(), 0u);
            }
            kp_off.push_back((uint32_t)hkx.size());
        }
        if (cloud) { ismhip_cloud_destroy(ctx, cloud); cloud = nullptr; }
        h2d(x, hx); h2d(y, hy); h2d(z, hz); h2d(nx, hnx); h2d(ny, hny); h2d(nz, hnz);
        has_color = with_color;
        if (with_color) h2d(rgba, hrgba);
        if (!dev_kp) {
            h2d(kx, hkx); h2d(ky, hky); h2d(kz, hkz);
            if (with_color) h2d(krgba, hkrgba);
        }
        finishBatch(dev_kp, cell, device_leaf);
    }
    // the point arrays of the batch are in HBM (x..rgba, pt_off): keypoints (device voxel grid when dev_kp) and the search surface
    void finishBatch(bool dev_kp, float cell, float device_leaf) {
        const bool with_color = has_color;
        if (cloud) { ismhip_cloud_destroy(ctx, cloud); cloud = nullptr; }
        if (dev_kp) {
            // KeypointsVoxelGrid::iComputeKeypoints on the device: centroids stay in HBM, only the per-object counts come back
            const size_t n_pts = pt_off.back();
            kx.reserve(std::max<size_t>(n_pts, 1) * 4); ky.reserve(std::max<size_t>(n_pts, 1) * 4); kz.reserve(std::max<size_t>(n_pts, 1) * 4);
            if (with_color) krgba.reserve(std::max<size_t>(n_pts, 1) * 4);
            kp_off.assign((size_t)n_obj + 1, 0);
            check(ismhip_voxel_keypoints(ctx, n_obj, pt_off.data(), x.as<float>(), y.as<float>(), z.as<float>(), with_color ? rgba.as<uint32_t>() : nullptr,
                                         device_leaf, (uint32_t)n_pts, kx.as<float>(), ky.as<float>(), kz.as<float>(),
                                         with_color ? krgba.as<uint32_t>() : nullptr, kp_off.data()), "ismhip_voxel_keypoints");
        }
        check(ismhip_cloud_create(ctx, n_obj, pt_off.data(), x.as<float>(), y.as<float>(), z.as<float>(), nx.as<float>(), ny.as<float>(),
                                  nz.as<float>(), with_color ? rgba.as<uint32_t>() : nullptr, cell, &cloud), "ismhip_cloud_create");
    }
};

// ---------------------------------------------------------------------------------------------------------------
// JSONObject (utils/json_object.cpp)
// ---------------------------------------------------------------------------------------------------------------
JSONObject::JSONObject() {}
JSONObject::~JSONObject() { for (auto* p : m_params) delete p; }

Json JSONObject::configToJson() const {        // json_object.cpp:180-205
    Json object = Json::object();
    if (!getType().empty()) object["Type"] = Json::of(getType());
    Json params = Json::object();
    for (auto* p : m_params) params[p->name] = p->toJson();
    if (!params.obj.empty()) object["Parameters"] = params;
    Json children = iChildConfigsToJson();
    if (children.isObject() && !children.obj.empty()) object["Children"] = children;
    return object;
}

bool JSONObject::configFromJson(const Json& object) {   // json_object.cpp:207-240
    if (!object.isObject()) return false;
    const Json* params = object.find("Parameters");
    for (auto* p : m_params) p->fromJson(params && params->isObject() ? params->find(p->name) : nullptr);
    iPostInitConfig();
    static const Json empty = Json::object();
    const Json* children = object.find("Children");
    return iChildConfigsFromJson(children ? *children : empty);
}

static std::string dirOf(const std::string& f) { size_t p = f.find_last_of('/'); return p == std::string::npos ? std::string() : f.substr(0, p + 1); }
static std::string baseOf(const std::string& f) { size_t p = f.find_last_of('/'); return p == std::string::npos ? f : f.substr(p + 1); }

bool JSONObject::writeObject(std::string file) {
    std::string data = file;
    size_t p = data.find_last_of('.');
    if (p != std::string::npos && p > data.find_last_of('/') + 0) data = data.substr(0, p);
    return writeObject(file, data + ".ismd");
}
bool JSONObject::writeObject(std::string file, std::string fileData) {     // json_object.cpp:50-95
    Json root = Json::object();
    root["ObjectConfig"] = configToJson();
    root["ObjectData"] = Json::of(baseOf(fileData));
    std::ofstream cfg(file);
    if (!cfg) { LOG_ERROR("could not write file: " << file); return false; }
    cfg << root.dump(3) << std::endl;
    std::ofstream data(fileData, std::ios::binary);
    if (!data) { LOG_ERROR("could not write file: " << fileData); return false; }
    iSaveData(data);
    return true;
}
bool JSONObject::readObject(std::string file, bool training) {            // json_object.cpp:97-178
    std::ifstream in(file);
    if (!in) { LOG_ERROR("could not read file: " << file); return false; }
    std::stringstream ss; ss << in.rdbuf();
    Json root;
    try { root = Json::parse(ss.str()); } catch (const std::exception& e) { throw JSONException(std::string("could not parse ") + file + ": " + e.what()); }
    const Json* cfg = root.find("ObjectConfig");
    if (!cfg) throw JSONException("no ObjectConfig in " + file);
    m_input_config_file = file;
    if (!configFromJson(*cfg)) return false;
    const Json* data = root.find("ObjectData");
    if (!training && data && data->type == Json::String && !data->str.empty()) {
        std::ifstream d(dirOf(file) + data->str, std::ios::binary);
        if (!d) { LOG_ERROR("could not read data file: " << dirOf(file) + data->str); return false; }
        return iLoadData(d);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Keypoints: pcl::VoxelGrid centroids (keypoints/keypoints_voxel_grid.cpp:30-46). Host side:
// the step BEFORE the hot path.
// ---------------------------------------------------------------------------------------------------------------
KeypointsVoxelGrid::KeypointsVoxelGrid() { addParameter(m_leafSize, "LeafSize", 0.1f); }

KeypointSet KeypointsVoxelGrid::iComputeKeypoints(const PointCloud& c) const {
    KeypointSet out;
    const size_t n = c.size();
    if (n == 0) return out;
    float mn[3] = {c.x[0], c.y[0], c.z[0]}, mx[3] = {c.x[0], c.y[0], c.z[0]};
    for (size_t i = 0; i < n; ++i) {
        mn[0] = std::min(mn[0], c.x[i]); mx[0] = std::max(mx[0], c.x[i]);
        mn[1] = std::min(mn[1], c.y[i]); mx[1] = std::max(mx[1], c.y[i]);
        mn[2] = std::min(mn[2], c.z[i]); mx[2] = std::max(mx[2], c.z[i]);
    }
    const float inv = 1.0f / m_leafSize;
    int minb[3], divb[3];
    for (int a = 0; a < 3; ++a) { minb[a] = (int)std::floor(mn[a] * inv); divb[a] = (int)std::floor(mx[a] * inv) - minb[a] + 1; }
    const int64_t mul1 = divb[0], mul2 = (int64_t)divb[0] * divb[1];
    std::vector<std::pair<int64_t, uint32_t>> iv(n);
    for (size_t i = 0; i < n; ++i) {
        const int64_t i0 = (int64_t)(std::floor(c.x[i] * inv) - (float)minb[0]);
        const int64_t i1 = (int64_t)(std::floor(c.y[i] * inv) - (float)minb[1]);
        const int64_t i2 = (int64_t)(std::floor(c.z[i] * inv) - (float)minb[2]);
        iv[i] = {i0 + i1 * mul1 + i2 * mul2, (uint32_t)i};
    }
    std::stable_sort(iv.begin(), iv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    const bool color = c.rgba.size() == n;
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        float s[3] = {0, 0, 0}, col[3] = {0, 0, 0};
        while (j < n && iv[j].first == iv[i].first) {
            const uint32_t p = iv[j].second;
            s[0] += c.x[p]; s[1] += c.y[p]; s[2] += c.z[p];
            if (color) { col[0] += (float)((c.rgba[p] >> 16) & 0xff); col[1] += (float)((c.rgba[p] >> 8) & 0xff); col[2] += (float)(c.rgba[p] & 0xff); }
            ++j;
        }
        const float cnt = (float)(j - i);
        out.x.push_back(s[0] / cnt); out.y.push_back(s[1] / cnt); out.z.push_back(s[2] / cnt);
        if (color) out.rgba.push_back(((uint32_t)(uint8_t)(col[0] / cnt) << 16) | ((uint32_t)(uint8_t)(col[1] / cnt) << 8) | (uint32_t)(uint8_t)(col[2] / cnt));
        i = j;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// Features (features/features.cpp)
// ---------------------------------------------------------------------------------------------------------------
Features::Features() : m_numThreads(0) {
    addParameter(m_referenceFrameRadius, "ReferenceFrameRadius", 0.2f);
    addParameter(m_referenceFrameType, "ReferenceFrameType", std::string("SHOT"));
}

std::shared_ptr<DeviceFeatures> Features::operator()(DeviceSession& s) const {   // features.cpp:40-116
    if (m_referenceFrameType != "SHOT") {
        if (m_referenceFrameType == "BOARD" || m_referenceFrameType == "FLARE" || m_referenceFrameType == "SHOTNA")
            throw RuntimeException("reference frame type \"" + m_referenceFrameType + "\" is not built on the MI355X path (only \"SHOT\")");
        throw BadParamExceptionType<std::string>("invalid reference frame type", m_referenceFrameType);   // features.cpp:178
    }
    const uint32_t nkp = s.kp_off.back();
    const int D = getDescriptorLength();
    auto f = std::make_shared<DeviceFeatures>();
    f->dim = D;
    f->off.assign(s.n_obj + 1, 0);
    if (nkp == 0) return f;
    LOG_INFO("computing reference frames");
    s.raw_lrf.reserve((size_t)nkp * 9 * 4); s.raw_desc.reserve((size_t)nkp * D * 4); s.raw_cnt.reserve((size_t)nkp * 4);
    s.check(ismhip_shot_lrf(s.ctx, s.cloud, s.kp_off.data(), s.kx.as<float>(), s.ky.as<float>(), s.kz.as<float>(), m_referenceFrameRadius,
                            s.raw_lrf.as<float>()), "ismhip_shot_lrf");
    LOG_INFO("computing descriptors at keypoint positions");
    iComputeDescriptors(s, s.raw_lrf.as<float>(), s.raw_desc.as<float>(), s.raw_cnt.as<uint32_t>());
    // invalid frames and NaN descriptors are dropped, order preserved (features.cpp:66-76, implicit_shape_model.cpp:1276-1308)
    f->desc.reserve((size_t)nkp * D * 4); f->lrf.reserve((size_t)nkp * 9 * 4);
    f->kx.reserve((size_t)nkp * 4); f->ky.reserve((size_t)nkp * 4); f->kz.reserve((size_t)nkp * 4); f->src.reserve((size_t)nkp * 4);
    int all_kept = 0;
    s.check(ismhip_compact_descriptor_rows(s.ctx, s.n_obj, s.kp_off.data(), D, s.raw_desc.as<float>(), s.raw_lrf.as<float>(), s.kx.as<float>(),
                                           s.ky.as<float>(), s.kz.as<float>(), f->desc.as<float>(), f->lrf.as<float>(), f->kx.as<float>(),
                                           f->ky.as<float>(), f->kz.as<float>(), f->src.as<uint32_t>(), f->off.data(), &all_kept),
            "ismhip_compact_descriptor_rows");
    if (all_kept) {
        // nothing was dropped and nothing was copied: the raw matrices become the features, the keypoints are duplicated (12 bytes each)
        f->desc.swap(s.raw_desc); f->lrf.swap(s.raw_lrf);
        s.sync();
        if (nkp && (hipMemcpy(f->kx.p, s.kx.p, (size_t)nkp * 4, hipMemcpyDeviceToDevice) != hipSuccess ||
                    hipMemcpy(f->ky.p, s.ky.p, (size_t)nkp * 4, hipMemcpyDeviceToDevice) != hipSuccess ||
                    hipMemcpy(f->kz.p, s.kz.p, (size_t)nkp * 4, hipMemcpyDeviceToDevice) != hipSuccess))
            throw RuntimeException("hipMemcpy D2D failed");
    }
    f->n = f->off.back();
    if (f->n < nkp) LOG_WARN("discarded " << (nkp - f->n) << " keypoint(s) with invalid reference frame or NaN descriptor");
    LOG_INFO("obtained " << f->n << " " << getType() << " descriptors");
    return f;
}

FeaturesSHOT::FeaturesSHOT() { addParameter(m_radius, "Radius", 0.1f); }
FeaturesCSHOT::FeaturesCSHOT() { addParameter(m_radius, "Radius", 0.1f); }
FeaturesFPFH::FeaturesFPFH() { addParameter(m_radius, "Radius", 0.1f); }

void FeaturesSHOT::iComputeDescriptors(DeviceSession& s, const float* lrf9, float* desc_out, uint32_t* counts_out) const {   // features_shot.cpp:28-81
    s.check(ismhip_shot352(s.ctx, s.cloud, s.kp_off.data(), s.kx.as<float>(), s.ky.as<float>(), s.kz.as<float>(), lrf9, m_radius, desc_out, counts_out),
            "ismhip_shot352");
}
void FeaturesCSHOT::iComputeDescriptors(DeviceSession& s, const float* lrf9, float* desc_out, uint32_t* counts_out) const {  // features_cshot.cpp:28-103
    if (!s.has_color) throw RuntimeException("CSHOT needs coloured point clouds");
    s.check(ismhip_cshot1344(s.ctx, s.cloud, s.kp_off.data(), s.kx.as<float>(), s.ky.as<float>(), s.kz.as<float>(), s.krgba.as<uint32_t>(), lrf9,
                             m_radius, desc_out, counts_out), "ismhip_cshot1344");
}
void FeaturesFPFH::iComputeDescriptors(DeviceSession& s, const float*, float* desc_out, uint32_t* counts_out) const {        // features_fpfh.cpp:27-72
    s.check(ismhip_fpfh33(s.ctx, s.cloud, s.kp_off.data(), s.kx.as<float>(), s.ky.as<float>(), s.kz.as<float>(), m_radius, desc_out, counts_out),
            "ismhip_fpfh33");
}

// ---------------------------------------------------------------------------------------------------------------
// activation strategy + codebook
// ---------------------------------------------------------------------------------------------------------------
ActivationStrategy::ActivationStrategy() {       // activation_strategy.cpp:16-22
    addParameter(m_use_distance_ratio, "UseDistanceRatio", false);
    addParameter(m_distance_ratio_threshold, "DistanceRatioThreshold", 0.95f);
}
ActivationStrategyKNN::ActivationStrategyKNN() { addParameter(m_k, "K", 1); }   // activation_strategy_knn.cpp:19

Codebook::Codebook() {                           // codebook.cpp:28-41
    m_activationStrategy.reset(new ActivationStrategyKNN());
    addParameter(m_useClassWeight, "UseClassWeight", false);
    addParameter(m_useVoteWeight, "UseVoteWeight", false);
    addParameter(m_useMatchingWeight, "UseMatchingWeight", false);
    addParameter(m_useCodewordWeight, "UseCodewordWeight", false);
    addParameter(m_use_partial_shot, "UsePartialShot", false);
    addParameter(m_partial_shot_type, "PartialShotType", std::string("front"));
    addParameter(m_use_random_codebook, "UseRandomCodebook", false);
    addParameter(m_random_codebook_factor, "RandomCodebookFactor", 1.0f);
}
Codebook::~Codebook() {
    if (m_dev && m_dev_session) ismhip_codebook_destroy(m_dev_session->ctx, m_dev);
}
Json Codebook::iChildConfigsToJson() const {
    Json c = Json::object();
    if (m_activationStrategy) c["ActivationStrategy"] = m_activationStrategy->configToJson();
    return c;
}
bool Codebook::iChildConfigsFromJson(const Json& c) {
    if (const Json* a = c.find("ActivationStrategy")) m_activationStrategy.reset(Factory<ActivationStrategy>::create(*a));
    return m_activationStrategy != nullptr;
}

// Codebook::getSignatureMask (codebook.cpp:952-1036) -> kept descriptor columns of SHOT-352 (11 per signature)
static std::vector<int32_t> partialShotColumns(const std::string& type) {
    std::vector<bool> m(32, false);
    auto rng = [&](int a, int b) { for (int i = a; i <= b; ++i) m[i] = true; };
    if (type == "front" || type == "dense_x") rng(8, 23);
    else if (type == "back" || type == "sparse_x") { rng(0, 7); rng(24, 31); }
    else if (type == "left" || type == "positive_y") rng(16, 31);
    else if (type == "right" || type == "negative_y") rng(0, 15);
    else if (type == "top" || type == "dense_z") { for (int i = 1; i < 32; i += 2) m[i] = true; }
    else if (type == "bottom" || type == "sparse_z") { for (int i = 0; i < 32; i += 2) m[i] = true; }
    else if (type == "dense_x_or_z") { rng(8, 23); for (int i = 1; i < 32; i += 2) m[i] = true; }
    else if (type == "dense_x_and_z") { for (int i = 9; i <= 23; i += 2) m[i] = true; }
    else if (type == "front_turn_left") rng(12, 27);
    else if (type == "front_turn_right") rng(4, 19);
    else { LOG_WARN("Unknown partial shot type: " << type << "! Using complete descriptor!"); m.assign(32, true); }
    std::vector<int32_t> cols;
    for (int sidx = 0; sidx < 32; ++sidx) if (m[sidx]) for (int j = 0; j < 11; ++j) cols.push_back(sidx * 11 + j);
    return cols;
}

void Codebook::upload(DeviceSession& s) const {
    if (!m_dirty && m_dev && m_dev_session == &s) return;
    if (m_dev && m_dev_session) ismhip_codebook_destroy(m_dev_session->ctx, m_dev);
    m_dev = nullptr; m_dev_session = &s;
    const CodebookData& d = m_data;
    if (d.numWords() == 0) return;
    std::vector<float> partial;
    const float* words = d.words.data(); int dim = d.dim;
    if (m_use_partial_shot) {
        // iLoadData builds the partial codewords (codebook.cpp:862-930). Plain SHOT only: with CSHOT the reference's loop leaks hist_size = 31
        // into the shape part of every feature after the first and produces descriptors of unequal length (:419, :458).
        if (d.dim != ISMHIP_SHOT_DIM) throw RuntimeException("UsePartialShot is built for SHOT-352 codebooks only");
        m_partial_cols = partialShotColumns(m_partial_shot_type);
        dim = (int)m_partial_cols.size();
        partial.resize((size_t)d.numWords() * dim);
        for (int w = 0; w < d.numWords(); ++w) for (int c = 0; c < dim; ++c) partial[(size_t)w * dim + c] = d.words[(size_t)w * d.dim + m_partial_cols[c]];
        words = partial.data();
    } else m_partial_cols.clear();
    s.check(ismhip_codebook_create(s.ctx, d.numWords(), dim, words, d.word_weight.empty() ? nullptr : d.word_weight.data(),
                                   d.vote_offsets.data(), d.vote_xyz.data(), d.vote_weight.empty() ? nullptr : d.vote_weight.data(),
                                   d.vote_class_weight.empty() ? nullptr : d.vote_class_weight.data(), d.vote_class.data(), d.vote_instance.data(),
                                   d.vote_bbox_quat.empty() ? nullptr : d.vote_bbox_quat.data(), d.vote_bbox_size.empty() ? nullptr : d.vote_bbox_size.data(),
                                   (int)d.class_sigma.size(), d.class_sigma.data(), &m_dev), "ismhip_codebook_create");
    if ((int)d.word_class.size() == d.numWords()) s.check(ismhip_codebook_set_word_class(s.ctx, m_dev, d.word_class.data()), "ismhip_codebook_set_word_class");
    m_dirty = false;
}

int ActivationStrategyKNN::activateKNN(DeviceSession& s, const ismhip_codebook* codewords, const DeviceFeatures& f, int metric,
                                       int32_t* idx_out, float* dist_out, const float* desc) const {     // activation_strategy_knn.h:41-126
    if (m_k > 16) throw RuntimeException("KNN activation with K > 16 is not built");
    if (f.n == 0) return m_k;
    if (!desc) desc = f.desc.as<float>();                        // desc: the (possibly partial, codebook.cpp:416-475) descriptors to match
    if (m_use_distance_ratio && m_is_detection && m_k == 1)
        s.check(ismhip_knn_ratio(s.ctx, codewords, metric, (int)f.n, desc, m_distance_ratio_threshold, idx_out, dist_out), "ismhip_knn_ratio");
    else
        s.check(ismhip_knn(s.ctx, codewords, metric, (int)f.n, desc, m_k, idx_out, dist_out), "ismhip_knn");
    return m_k;
}

ActivationStrategyKnnRule::ActivationStrategyKnnRule() { addParameter(m_k, "K", 3); m_is_detection = false; m_k = 3; }   // activation_strategy_knn_rule.cpp:16-22
int ActivationStrategyKnnRule::activateKNN(DeviceSession& s, const ismhip_codebook* codewords, const DeviceFeatures& f, int metric,
                                           int32_t* idx_out, float* dist_out, const float* desc) const {
    if (f.n == 0) return 1;
    if (!desc) desc = f.desc.as<float>();
    if (!m_is_detection) s.check(ismhip_knn(s.ctx, codewords, metric, (int)f.n, desc, 1, idx_out, dist_out), "ismhip_knn");
    else s.check(ismhip_knn_rule(s.ctx, codewords, metric, (int)f.n, desc, m_distance_ratio_threshold, idx_out, dist_out), "ismhip_knn_rule");
    return 1;
}

// FLANN functors on the host, used by the training statistics only (utils/distance.cpp:33-52)
// Utils::getRotQuaternion + matrix2Quat (utils.cpp:136-151, 342-380): rows of the matrix are the frame axes; out = (w, x, y, z)
static void hostRotQuaternion(const float* l, float* out) {
    const float m[3][3] = {{l[0], l[1], l[2]}, {l[3], l[4], l[5]}, {l[6], l[7], l[8]}};
    float q[4] = {0.f, 0.f, 0.f, 1.f};
    const float trace = m[0][0] + m[1][1] + m[2][2];
    float root;
    if (trace > 0.0f) {
        root = sqrtf(trace + 1.0f); q[3] = 0.5f * root; root = 0.5f / root;
        q[0] = (m[2][1] - m[1][2]) * root; q[1] = (m[0][2] - m[2][0]) * root; q[2] = (m[1][0] - m[0][1]) * root;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        root = sqrtf((float)((double)(m[i][i] - m[j][j] - m[k][k]) + 1.0)); q[i] = 0.5f * root; root = 0.5f / root;
        q[3] = (m[k][j] - m[j][k]) * root; q[j] = (m[j][i] + m[i][j]) * root; q[k] = (m[k][i] + m[i][k]) * root;
    }
    out[0] = q[3]; out[1] = q[0]; out[2] = q[1]; out[3] = q[2];
}
static float hostDistance(int metric, const float* a, const float* b, int n) {
    float result = 0.f;
    if (metric == ISMHIP_METRIC_CHI2) {
        for (int i = 0; i < n; ++i) { const float sum = a[i] + b[i]; if (sum > 0) { const float diff = a[i] - b[i]; result += diff * diff / sum; } }
        return result;
    }
    int i = 0;
    for (; i + 3 < n; i += 4) {
        const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; i < n; ++i) { const float d0 = a[i] - b[i]; result += d0 * d0; }
    return result;
}

void Codebook::activate(DeviceSession& s, const DeviceFeatures& f, const std::vector<unsigned>& feat_class, const std::vector<unsigned>& feat_instance,
                        const std::vector<unsigned>& feat_model, const std::vector<std::array<float, 3>>& feat_center,
                        const std::vector<std::array<float, 3>>& feat_bbox_size, int metric, int n_classes, const Clustering& clustering) {
    // codebook.cpp:64-368 with KNN / KNNRule activation; codewords = cluster centres (implicit_shape_model.cpp:445-475), or one per
    // training feature for Clustering "None" (clustering_none.cpp:25-35)
    const uint32_t n = f.n;
    const int D = f.dim;
    if (n == 0) throw RuntimeException("no training features");
    const ActivationStrategy* knn = m_activationStrategy.get();
    const bool is_knn = m_activationStrategy->getType() == "KNN";
    const int k = is_knn ? knn->getK() : 1;                      // KNNRule trains with plain 1-NN (activation_strategy_knn_rule.h:70-74)
    if (k > 16) throw RuntimeException("KNN activation with K > 16 is not built");
    // the whole of Codebook::activate runs on the device (ismhip_train_activate): self-kNN, class sigma^2, K = 1 clean-up,
    // vote CSR, computeWeights and the statistical class weights. Features must be class-major, as train() collects them.
    const bool clustered = clustering.hasCenters();
    const uint32_t n_cw = clustered ? (uint32_t)clustering.getNumCenters() : n;
    std::vector<uint32_t> cluster_size(n_cw, 1u);                // Codeword::m_numFeatures = clusters[i].size() (:453-473)
    if (clustered) { std::fill(cluster_size.begin(), cluster_size.end(), 0u); for (int ci : clustering.getClusterIndices()) if (ci >= 0 && (uint32_t)ci < n_cw) cluster_size[ci]++; }
    std::vector<float> words((size_t)n_cw * D), centers((size_t)n * 3), lrf, kx, ky, kz;
    if (hipMemcpy(words.data(), clustered ? clustering.getClusterCentersDevice() : f.desc.as<float>(), words.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        throw RuntimeException("hipMemcpy (codewords) failed");
    s.d2h(lrf, f.lrf, (size_t)n * 9); s.d2h(kx, f.kx, n); s.d2h(ky, f.ky, n); s.d2h(kz, f.kz, n);
    for (uint32_t i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) centers[(size_t)i * 3 + d] = feat_center[i][d];
    const bool clean_up = is_knn && k == 1;                     // codebook.cpp:201-224
    int32_t n_words = 0;
    std::vector<uint32_t> word_src(n_cw), vote_off((size_t)n_cw + 1), vote_feature((size_t)n * k);
    std::vector<float> vote_xyz((size_t)n * k * 3), vote_weight((size_t)n * k), vote_cw((size_t)n * k), sigma((size_t)std::max(1, n_classes));
    s.check(ismhip_train_activate(s.ctx, metric, (int)n, D, f.desc.as<float>(), f.lrf.as<float>(), f.kx.as<float>(), f.ky.as<float>(), f.kz.as<float>(),
                                  feat_class.data(), feat_model.data(), centers.data(), (int)n_cw, clustered ? clustering.getClusterCentersDevice() : nullptr,
                                  k, clean_up ? 1 : 0, std::max(1, n_classes), &n_words, word_src.data(),
                                  vote_off.data(), vote_feature.data(), vote_xyz.data(), vote_weight.data(), vote_cw.data(), sigma.data()), "ismhip_train_activate");
    std::vector<uint32_t> kept(word_src.begin(), word_src.begin() + n_words);
    std::vector<uint32_t> sel(kept.size());
    std::iota(sel.begin(), sel.end(), 0u);
    if (m_use_random_codebook && m_random_codebook_factor < 1.0f) {   // codebook.cpp:821-829, seeded instead of std::random_device
        std::mt19937 rng(0x5EED);
        std::uniform_int_distribution<uint32_t> distr(0, (uint32_t)kept.size());
        std::vector<uint32_t> sub;
        for (uint32_t e : sel) if (!(distr(rng) > m_random_codebook_factor * kept.size())) sub.push_back(e);
        sel.swap(sub);
    }
    CodebookData out; out.dim = D; out.class_sigma = sigma; out.vote_offsets.assign(1, 0u);
    for (uint32_t e : sel) {
        const uint32_t w = kept[e];
        out.words.insert(out.words.end(), words.begin() + (size_t)w * D, words.begin() + (size_t)(w + 1) * D);
        out.word_weight.push_back(1.0f);
        // Codeword(centre, clusters[i].size(), 1.0f, keypoint and class of FEATURE i) (implicit_shape_model.cpp:466-475): with a
        // clustered codebook that is the i-th feature, not a member of cluster i -- kept as the reference has it
        out.word_class.push_back(feat_class[w]);
        out.word_id.push_back((int32_t)w); out.word_num_features.push_back((int32_t)cluster_size[w]);
        out.word_keypoint.push_back(kx[w]); out.word_keypoint.push_back(ky[w]); out.word_keypoint.push_back(kz[w]);
        for (uint32_t v = vote_off[e]; v < vote_off[e + 1]; ++v) {
            const uint32_t fi = vote_feature[v];
            out.vote_xyz.push_back(vote_xyz[(size_t)v * 3]); out.vote_xyz.push_back(vote_xyz[(size_t)v * 3 + 1]); out.vote_xyz.push_back(vote_xyz[(size_t)v * 3 + 2]);
            out.vote_class.push_back(feat_class[fi]); out.vote_instance.push_back(feat_instance[fi]);
            out.vote_weight.push_back(vote_weight[v]); out.vote_class_weight.push_back(vote_cw[v]);
            // addCodeword (codeword_distribution.cpp:63-70): the box in the keypoint's frame, rotQuat = box.rotQuat * conj(q(LRF)); AABB boxes carry (1,0,0,0)
            float q[4]; hostRotQuaternion(&lrf[(size_t)fi * 9], q);
            out.vote_bbox_quat.push_back(q[0]); out.vote_bbox_quat.push_back(-q[1]); out.vote_bbox_quat.push_back(-q[2]); out.vote_bbox_quat.push_back(-q[3]);
            for (int d = 0; d < 3; ++d) out.vote_bbox_size.push_back(feat_bbox_size[fi][d]);
        }
        out.vote_offsets.push_back((uint32_t)out.vote_class.size());
    }
    LOG_INFO("Size of distribution at the end of training: " << sel.size());
    setData(out);
}

void Codebook::castVotes(DeviceSession& s, const DeviceFeatures& f, int metric, Voting& voting) const {   // codebook.cpp:403-555
    (void)voting;
    s.n_slots = 0; s.slot_off.assign(s.n_obj + 1, 0);
    if (isEmpty()) return;
    upload(s);
    m_activationStrategy->setIsDetection();
    const ActivationStrategy* knn = m_activationStrategy.get();
    if (knn->getK() > 16) throw RuntimeException("KNN activation with K > 16 is not built");
    const uint32_t n = f.n;
    if (n == 0) return;
    s.idx.reserve((size_t)n * 4 * 4); s.dist.reserve((size_t)n * 4 * 4);
    const float* qdesc = nullptr;
    DevBuf partial;
    if (!m_partial_cols.empty()) {                              // reduce every feature the way the codewords were reduced (codebook.cpp:416-475)
        if (f.dim != ISMHIP_SHOT_DIM) throw RuntimeException("UsePartialShot needs SHOT-352 features");
        partial.reserve((size_t)n * m_partial_cols.size() * 4);
        s.check(ismhip_gather_columns(s.ctx, (int)n, f.dim, f.desc.as<float>(), (int)m_partial_cols.size(), m_partial_cols.data(), partial.as<float>()), "ismhip_gather_columns");
        qdesc = partial.as<float>();
    }
    const int k = knn->activateKNN(s, m_dev, f, metric, s.idx.as<int32_t>(), s.dist.as<float>(), qdesc);
    if (qdesc) s.check(ismhip_sync(s.ctx), "ismhip_sync");     // the partial descriptors are released when this function returns
    const int maxv = ismhip_codebook_max_votes_per_word(m_dev);
    const size_t ns = (size_t)n * k * maxv;
    s.v_pos.reserve(ns * 12); s.v_w.reserve(ns * 4); s.v_cls.reserve(ns * 4); s.v_inst.reserve(ns * 4); s.v_cw.reserve(ns * 4); s.v_bs.reserve(ns * 12); s.v_bq.reserve(ns * 16);
    const uint32_t flags = (m_useClassWeight ? ISMHIP_W_CLASS : 0u) | (m_useVoteWeight ? ISMHIP_W_VOTE : 0u) |
                           (m_useMatchingWeight ? ISMHIP_W_MATCHING : 0u) | (m_useCodewordWeight ? ISMHIP_W_CODEWORD : 0u);
    s.check(ismhip_cast_votes(s.ctx, m_dev, flags, (int)n, f.lrf.as<float>(), f.kx.as<float>(), f.ky.as<float>(), f.kz.as<float>(), k, s.idx.as<int32_t>(),
                              s.dist.as<float>(), s.v_pos.as<float>(), s.v_w.as<float>(), s.v_cls.as<int32_t>(), s.v_inst.as<int32_t>(), s.v_cw.as<int32_t>(),
                              s.v_bq.as<float>(), s.v_bs.as<float>()), "ismhip_cast_votes");
    s.n_slots = ns;
    for (int o = 0; o <= s.n_obj; ++o) s.slot_off[o] = f.off[o] * (uint32_t)(k * maxv);
    s.n_classes = (int)m_data.class_sigma.size();
}

std::string CodebookData::validate() const {
    const size_t nw = (size_t)numWords(), nv = vote_offsets.empty() ? 0 : vote_offsets.back();
    if (dim <= 0 || words.size() % (size_t)dim) return "descriptor length does not divide the word array";
    if (vote_offsets.size() != nw + 1 || vote_offsets[0] != 0) return "vote offsets do not match the number of codewords";
    for (size_t i = 0; i < nw; ++i) if (vote_offsets[i + 1] < vote_offsets[i]) return "vote offsets not monotone";
    if (vote_xyz.size() != nv * 3 || vote_class.size() != nv || vote_instance.size() != nv) return "vote arrays do not match the vote offsets";
    if (!vote_weight.empty() && vote_weight.size() != nv) return "vote weights do not match the vote offsets";
    if (!vote_class_weight.empty() && vote_class_weight.size() != nv) return "class weights do not match the vote offsets";
    if (!vote_bbox_quat.empty() && vote_bbox_quat.size() != nv * 4) return "bounding-box quaternions do not match the vote offsets";
    if (!vote_bbox_size.empty() && vote_bbox_size.size() != nv * 3) return "bounding-box sizes do not match the vote offsets";
    if (!word_weight.empty() && word_weight.size() != nw) return "word weights do not match the number of codewords";
    if (!word_class.empty() && word_class.size() != nw) return "word classes do not match the number of codewords";
    if (class_sigma.empty()) return "no class sigmas";
    for (uint32_t c : vote_class) if (c >= class_sigma.size()) return "vote class id beyond the class sigmas";
    return "";
}

// Codebook::iSaveData (codebook.cpp:739-761) -> CodewordDistribution::iSaveData (codeword_distribution.cpp:349-391) ->
// Codeword::iSaveData (codeword.cpp:71-83), field by field
void Codebook::save(BoostBinaryOArchive& oa) const {
    const CodebookData& d = m_data;
    const int nw = d.numWords();
    oa << nw;                                                                     // distribution_size
    for (int w = 0; w < nw; ++w) {
        // Codeword: m_id, m_numFeatures, m_weight, m_data, m_class_id, keypoint xyz
        oa << (d.word_id.empty() ? w : (int)d.word_id[w]) << (d.word_num_features.empty() ? 1 : (int)d.word_num_features[w])
           << (d.word_weight.empty() ? 1.0f : d.word_weight[w]);
        oa << std::vector<float>(d.words.begin() + (size_t)w * d.dim, d.words.begin() + (size_t)(w + 1) * d.dim);
        const uint32_t v0 = d.vote_offsets[w], v1 = d.vote_offsets[w + 1];
        oa << (int)(d.word_class.empty() ? (v1 > v0 ? d.vote_class[v0] : 0u) : d.word_class[w]);
        for (int k = 0; k < 3; ++k) oa << (d.word_keypoint.empty() ? 0.0f : d.word_keypoint[(size_t)w * 3 + k]);
        // distribution: votes, m_weights, m_class_ids, m_instance_ids, m_classWeights (std::map: ascending class), bounding boxes
        oa << (int)(v1 - v0);
        for (uint32_t v = v0; v < v1; ++v) oa << d.vote_xyz[(size_t)v * 3] << d.vote_xyz[(size_t)v * 3 + 1] << d.vote_xyz[(size_t)v * 3 + 2];
        oa << (d.vote_weight.empty() ? std::vector<float>(v1 - v0, 1.0f) : std::vector<float>(d.vote_weight.begin() + v0, d.vote_weight.begin() + v1));
        oa << std::vector<unsigned>(d.vote_class.begin() + v0, d.vote_class.begin() + v1);
        oa << std::vector<unsigned>(d.vote_instance.begin() + v0, d.vote_instance.begin() + v1);
        std::map<unsigned, float> cw;
        for (uint32_t v = v0; v < v1; ++v) cw[d.vote_class[v]] = d.vote_class_weight.empty() ? 1.0f : d.vote_class_weight[v];
        oa << (int)cw.size();
        for (auto& kv : cw) oa << (int)kv.first << kv.second;
        oa << (int)(v1 - v0);
        for (uint32_t v = v0; v < v1; ++v) {
            for (int k = 0; k < 4; ++k) oa << (d.vote_bbox_quat.empty() ? (k == 0 ? 1.0f : 0.0f) : d.vote_bbox_quat[(size_t)v * 4 + k]);
            for (int k = 0; k < 3; ++k) oa << (d.vote_bbox_size.empty() ? 0.0f : d.vote_bbox_size[(size_t)v * 3 + k]);
        }
    }
    // m_classSigmas (std::map<unsigned, float>): only classes that were trained have an entry
    int n_sig = 0;
    for (float sg : d.class_sigma) if (sg == sg) ++n_sig;      // NaN marks a class id that was never trained: no map entry
    oa << n_sig;
    for (size_t c = 0; c < d.class_sigma.size(); ++c) if (d.class_sigma[c] == d.class_sigma[c]) oa << (int)c << d.class_sigma[c];
    // m_activationStrategy->saveData: nothing (json_object.cpp:256-259)
}

bool Codebook::load(BoostBinaryIArchive& ia) {
    CodebookData d;
    int nw = 0; ia >> nw;
    if (!ia.ok() || nw < 0 || !ia.plausible((uint64_t)nw, 60)) return false;
    LOG_INFO("Loading codebook with size: " << nw);
    std::mt19937 rng(0x5EED);                                  // UseRandomCodebook (codebook.cpp:821-829), seeded instead of std::random_device
    std::uniform_int_distribution<int> pick(0, nw);
    std::map<unsigned, float> sig_map;
    for (int w = 0; w < nw && ia.ok(); ++w) {
        int id = 0, nf = 0, cls = 0; float weight = 0.f, kp[3] = {0, 0, 0};
        std::vector<float> data, weights; std::vector<unsigned> class_ids, instance_ids;
        ia >> id >> nf >> weight >> data >> cls >> kp[0] >> kp[1] >> kp[2];
        int votes = 0; ia >> votes;
        if (!ia.ok() || votes < 0 || !ia.plausible((uint64_t)votes, 12)) return false;
        std::vector<float> vxyz((size_t)votes * 3);
        for (float& x : vxyz) ia >> x;
        ia >> weights >> class_ids >> instance_ids;
        int ncw = 0; ia >> ncw;
        if (!ia.ok() || ncw < 0 || !ia.plausible((uint64_t)ncw, 8)) return false;
        std::map<unsigned, float> cw;
        for (int i = 0; i < ncw; ++i) { int c = 0; float x = 0.f; ia >> c >> x; cw[(unsigned)c] = x; }
        int nbb = 0; ia >> nbb;
        if (!ia.ok() || nbb < 0 || !ia.plausible((uint64_t)nbb, 28)) return false;
        std::vector<float> bq((size_t)nbb * 4), bs((size_t)nbb * 3);
        for (int i = 0; i < nbb; ++i) { for (int k = 0; k < 4; ++k) ia >> bq[(size_t)i * 4 + k]; for (int k = 0; k < 3; ++k) ia >> bs[(size_t)i * 3 + k]; }
        if (!ia.ok()) return false;
        if ((int)weights.size() != votes || (int)class_ids.size() != votes || (int)instance_ids.size() != votes || nbb != votes) { ia.fail("codeword distribution arrays of unequal length"); return false; }
        if (d.dim == 0) d.dim = (int)data.size();
        if ((int)data.size() != d.dim || d.dim == 0) { ia.fail("codewords of different descriptor length"); return false; }
        if (m_use_random_codebook && pick(rng) > m_random_codebook_factor * nw) continue;      // skip features while loading (:821-829)
        d.words.insert(d.words.end(), data.begin(), data.end());
        d.word_id.push_back(id); d.word_num_features.push_back(nf); d.word_weight.push_back(weight); d.word_class.push_back((uint32_t)cls);
        d.word_keypoint.insert(d.word_keypoint.end(), kp, kp + 3);
        d.vote_xyz.insert(d.vote_xyz.end(), vxyz.begin(), vxyz.end());
        d.vote_weight.insert(d.vote_weight.end(), weights.begin(), weights.end());
        d.vote_class.insert(d.vote_class.end(), class_ids.begin(), class_ids.end());
        d.vote_instance.insert(d.vote_instance.end(), instance_ids.begin(), instance_ids.end());
        for (unsigned c : class_ids) { auto it = cw.find(c); d.vote_class_weight.push_back(it == cw.end() ? 1.0f : it->second); }   // castVotes: 1 + warning when missing (:97-104)
        d.vote_bbox_quat.insert(d.vote_bbox_quat.end(), bq.begin(), bq.end()); d.vote_bbox_size.insert(d.vote_bbox_size.end(), bs.begin(), bs.end());
        d.vote_offsets.push_back((uint32_t)d.vote_class.size());
    }
    int n_sig = 0; ia >> n_sig;
    if (!ia.ok() || n_sig < 0 || !ia.plausible((uint64_t)n_sig, 8)) return false;
    unsigned max_class = 0;
    for (int i = 0; i < n_sig; ++i) { int c = 0; float sg = 0.f; ia >> c >> sg; if (c < 0 || c > (1 << 20)) { ia.fail("class id out of range"); return false; } sig_map[(unsigned)c] = sg; max_class = std::max(max_class, (unsigned)c); }
    for (unsigned c : d.vote_class) max_class = std::max(max_class, c);
    if (max_class > (1u << 20)) { ia.fail("class id out of range"); return false; }
    d.class_sigma.assign((size_t)max_class + 1, 1.0f);          // castVotes uses sigma 1 (+ warning) for a class without an entry (:108-117)
    for (auto& kv : sig_map) d.class_sigma[kv.first] = kv.second;
    if (!ia.ok()) return false;
    const std::string why = d.validate();
    if (!why.empty()) { ia.fail("inconsistent codebook: " + why); return false; }
    if (m_use_random_codebook) LOG_INFO("Reduced codebook size: " << d.numWords());
    setData(d);
    return true;
}

// Voting::forwardBoxesAndRadii (voting.cpp:496-551)
void Voting::forwardBoxesAndRadii(const std::map<unsigned, std::vector<std::array<float, 3>>>& box_sizes, const std::map<unsigned, std::vector<float>>& object_radii) {
    m_dimensions_map.clear(); m_variance_map.clear();
    for (auto& it : box_sizes) {
        const unsigned classId = it.first;
        float median_box_dim = 0, median_box_dim_squared = 0;
        for (auto& size : it.second) {
            const float mx = std::max(size[0], std::max(size[1], size[2])), mn = std::min(size[0], std::min(size[1], size[2]));
            float med = size[0];                               // "find the other value" (:512-520)
            for (int i = 1; i < 3; ++i) if (med == mx || med == mn) med = size[i];
            median_box_dim += med; median_box_dim_squared += med * med;
        }
        float class_radii = 0, class_radii_squared = 0;
        auto rit = object_radii.find(classId);
        if (rit != object_radii.end()) for (float r : rit->second) { class_radii += r; class_radii_squared += r * r; }
        const float n = (float)it.second.size();
        median_box_dim /= n; median_box_dim_squared /= n; class_radii /= n; class_radii_squared /= n;
        m_dimensions_map[classId] = {class_radii, median_box_dim};
        m_variance_map[classId] = {class_radii_squared - class_radii * class_radii, median_box_dim_squared - median_box_dim * median_box_dim};
    }
}
// MaximaHandler::getSearchDistForClass (maxima_handler.cpp:509-521)
std::vector<float> Voting::searchDistPerClass(float radius, int n_classes) const {
    if (m_radiusType == "Config" || m_radiusType == "Fixed") return {};
    const bool first = m_radiusType == "FirstDim" || m_radiusType == "ObjectRadius", second = m_radiusType == "SecondDim" || m_radiusType == "BoundingBoxMedian";
    if (!first && !second) { LOG_ERROR("Invalid radius type: " << m_radiusType << "! Using config value instead."); return {}; }
    std::vector<float> out((size_t)n_classes, radius);
    for (int c = 0; c < n_classes; ++c) {
        auto it = m_dimensions_map.find((unsigned)c);
        if (it == m_dimensions_map.end()) continue;            // a class that was never trained casts no votes either
        out[c] = (first ? it->second.first : it->second.second) * m_radiusFactor;
    }
    return out;
}
void Voting::save(BoostBinaryOArchive& oa) const {
    oa << (unsigned)m_dimensions_map.size();
    for (auto& it : m_dimensions_map) oa << it.first << it.second.first << it.second.second;
    oa << (unsigned)m_variance_map.size();
    for (auto& it : m_variance_map) oa << it.first << it.second.first << it.second.second;
    oa << 0u;                                                  // global features: none (out of scope)
}
bool Voting::load(BoostBinaryIArchive& ia) {
    m_dimensions_map.clear(); m_variance_map.clear();
    unsigned n = 0; ia >> n;
    if (!ia.plausible(n, 12)) return false;
    for (unsigned i = 0; i < n; ++i) { unsigned c = 0; float a = 0, b = 0; ia >> c >> a >> b; m_dimensions_map[c] = {a, b}; }
    ia >> n;
    if (!ia.plausible(n, 12)) return false;
    for (unsigned i = 0; i < n; ++i) { unsigned c = 0; float a = 0, b = 0; ia >> c >> a >> b; m_variance_map[c] = {a, b}; }
    // global features must be deserialised completely even when unused (voting.cpp:652-705); they are read and dropped
    unsigned n_glob = 0; ia >> n_glob;
    if (!ia.plausible(n_glob, 8)) return false;
    for (unsigned i = 0; i < n_glob && ia.ok(); ++i) {
        unsigned classId = 0, clouds = 0; ia >> classId >> clouds;
        if (!ia.plausible(clouds, 4)) return false;
        for (unsigned j = 0; j < clouds && ia.ok(); ++j) {
            unsigned feats = 0; ia >> feats;
            if (!ia.plausible(feats, 52)) return false;
            for (unsigned k = 0; k < feats && ia.ok(); ++k) {
                float rf; for (int r = 0; r < 9; ++r) ia >> rf;
                std::vector<float> desc; float radius; unsigned inst; ia >> desc >> radius >> inst;
            }
        }
    }
    return ia.ok();
}

// ---------------------------------------------------------------------------------------------------------------
// Voting (voting/voting.cpp, voting_mean_shift.cpp)
// ---------------------------------------------------------------------------------------------------------------
Voting::Voting() {                                // voting.cpp:28-50 (hot-path subset; the global-feature / RANSAC keys are accepted and must stay off)
    addParameter(m_minThreshold, "MinThreshold", 0.0f);
    addParameter(m_minVotesThreshold, "MinVotesThreshold", 1);
    addParameter(m_bestK, "BestK", -1);
    addParameter(m_averageRotation, "AverageRotation", false);
    addParameter(m_radiusType, "BinOrBandwidthType", std::string("Config"));
    addParameter(m_radiusFactor, "BinOrBandwidthFactor", 1.0f);
    addParameter(m_max_filter_type, "MaxFilterType", std::string("None"));
    addParameter(m_max_type_param, "SingleObjectMaxType", std::string("Default"));
    addParameter(m_single_object_mode, "SingleObjectMode", false);
    addParameter(m_use_global_features, "UseGlobalFeatures", false);
    addParameter(m_vote_filtering_with_ransac, "RansacVoteFiltering", false);
}
void Voting::clear() {}
std::vector<std::vector<VotingMaximum>> Voting::findMaxima(DeviceSession& s) {
    if (m_use_global_features) throw RuntimeException("UseGlobalFeatures is out of scope of the MI355X path (SURVEY §2 row 10)");
    if (m_vote_filtering_with_ransac) throw RuntimeException("RansacVoteFiltering is not built on the MI355X path");
    if (m_max_filter_type != "None" && m_max_filter_type != "Simple" && m_max_filter_type != "Merge")
        LOG_ERROR("Invalid maxima filter type specified: " << m_max_filter_type << "! No filtering is performed!");            // maxima_handler.cpp:292-295
    if (m_max_type_param != "None" && m_max_type_param != "Default" && m_max_type_param != "BandwidthVotes" && m_max_type_param != "VotingSpaceVotes" &&
        m_max_type_param != "ModelRadiusVotes")
        LOG_WARN("Invalid single object maximum type: " << m_max_type_param << "! Using default instead.");                   // maxima_handler.h:53-57
    std::vector<std::vector<VotingMaximum>> out(s.n_obj);
    if (s.n_slots == 0) return out;
    if (singleObjectMaxType() != ISMHIP_SOM_MEANSHIFT) {       // voting_mean_shift.cpp:124-157: the query point is the centroid of the object's cloud
        s.obj_cen.reserve((size_t)s.n_obj * 12); s.obj_rad.reserve((size_t)s.n_obj * 4);
        s.check(ismhip_cloud_centroids(s.ctx, s.cloud, s.obj_cen.as<float>()), "ismhip_cloud_centroids");
        s.check(ismhip_cloud_radii(s.ctx, s.cloud, s.obj_cen.as<float>(), s.obj_rad.as<float>()), "ismhip_cloud_radii");
    }
    iFindMaxima(s, out);
    return out;
}
int Voting::singleObjectMaxType() const {                      // maxima_handler.h:42-58; only consulted in single-object mode (voting_mean_shift.cpp:80)
    if (!m_single_object_mode) return ISMHIP_SOM_MEANSHIFT;
    if (m_max_type_param == "BandwidthVotes") return ISMHIP_SOM_BANDWIDTH;
    if (m_max_type_param == "VotingSpaceVotes") return ISMHIP_SOM_COMPLETE_VOTING_SPACE;
    if (m_max_type_param == "ModelRadiusVotes") return ISMHIP_SOM_MODEL_RADIUS;
    return ISMHIP_SOM_MEANSHIFT;
}
int Voting::maxFilter() const {                                // voting.cpp:262-268: no inter-class filter in single-object mode
    if (m_single_object_mode) return ISMHIP_MAXFILTER_NONE;
    return m_max_filter_type == "Simple" ? ISMHIP_MAXFILTER_SIMPLE : (m_max_filter_type == "Merge" ? ISMHIP_MAXFILTER_MERGE : ISMHIP_MAXFILTER_NONE);
}

// M = maxima per object the buffers hold. The reference returns every maximum (voting.cpp:236-272), the device call a fixed capacity:
// the host asks for 32 and, when some object fills them all, again for 1024 = the kernels' own per-object limit, beyond which
// ismhip_sync reports the truncation.
struct Voting::MaximaBuffers { DevBuf n_max, pos, w, cls, inst, iw, bs, bq, nv, score; int M = 32; void reserve(int n_obj, int C); };
VotingMeanShift::VotingMeanShift() {              // voting_mean_shift.cpp:20-27
    addParameter(m_bandwidth, "Bandwidth", 0.2f);
    addParameter(m_threshold, "Threshold", 1e-3f);
    addParameter(m_maxIter, "MaxIter", 1000);
    addParameter(m_kernel, "Kernel", std::string("Gaussian"));
    addParameter(m_maxima_suppression_type, "MaximaSuppression", std::string("Average"));
}
void VotingMeanShift::iFindMaxima(DeviceSession& s, std::vector<std::vector<VotingMaximum>>& out) {
    const int C = std::max(1, s.n_classes);
    for (int M : {32, 1024}) {
    ismhip_maxima_params P{};
    const std::vector<float> class_bw = searchDistPerClass(m_bandwidth, C);      // voting_mean_shift.cpp:46-49
    P.n_classes = C; P.class_bandwidth_h = class_bw.empty() ? nullptr : class_bw.data(); P.bandwidth = m_bandwidth; P.threshold = m_threshold; P.max_iter = m_maxIter;
    P.kernel = m_kernel == "Uniform" ? ISMHIP_KERNEL_UNIFORM : ISMHIP_KERNEL_GAUSSIAN;
    P.suppression = m_maxima_suppression_type == "Average" ? ISMHIP_SUPPRESS_AVERAGE : (m_maxima_suppression_type == "Suppress" ? ISMHIP_SUPPRESS_SUPPRESS : ISMHIP_SUPPRESS_NONE);
    P.min_votes_threshold = m_minVotesThreshold; P.min_threshold = m_minThreshold; P.best_k = m_bestK; P.max_maxima = M;
    P.max_filter = maxFilter();
    MaximaBuffers B; B.M = M; B.reserve(s.n_obj, C);
    if (m_averageRotation) { P.vote_bbox_quat = s.v_bq.as<float>(); P.max_bbox_quat_out = B.bq.as<float>(); }              // voting.cpp:210-215
    P.single_object_max_type = singleObjectMaxType(); P.object_centroid = s.obj_cen.as<float>(); P.object_radius = s.obj_rad.as<float>();
    s.check(ismhip_find_maxima(s.ctx, s.n_obj, s.slot_off.data(), s.v_pos.as<float>(), s.v_w.as<float>(), s.v_cls.as<int32_t>(), s.v_inst.as<int32_t>(),
                               s.v_bs.as<float>(), &P, B.n_max.as<int32_t>(), B.pos.as<float>(), B.w.as<float>(), B.cls.as<int32_t>(), B.inst.as<int32_t>(),
                               B.iw.as<float>(), B.bs.as<float>(), B.nv.as<int32_t>(), B.score.as<float>()), "ismhip_find_maxima");
    if (collectMaxima(s, B, out, m_averageRotation) || M == 1024) break;
    }
}
void Voting::MaximaBuffers::reserve(int n_obj, int C) {
    const size_t t = (size_t)n_obj * M;
    n_max.reserve((size_t)n_obj * 4); pos.reserve(t * 12); w.reserve(t * 4); cls.reserve(t * 4); inst.reserve(t * 4); iw.reserve(t * 4); bs.reserve(t * 12);
    nv.reserve(t * 4); score.reserve((size_t)n_obj * C * 4); bq.reserve(t * 16);
}
// returns false when some object filled all M slots (the caller asks again with more room)
bool Voting::collectMaxima(DeviceSession& s, MaximaBuffers& b, std::vector<std::vector<VotingMaximum>>& out, bool with_quat) {
    const int M = b.M; const size_t t = (size_t)s.n_obj * M;
    std::vector<int32_t> hn, hcls, hinst, hnv; std::vector<float> hpos, hw, hiw, hbs, hbq;
    s.d2h(hn, b.n_max, s.n_obj);
    for (int o = 0; o < s.n_obj; ++o) if (hn[o] >= M && M < 1024) return false;
    if (with_quat) s.d2h(hbq, b.bq, t * 4);
    s.d2h(hpos, b.pos, t * 3); s.d2h(hw, b.w, t); s.d2h(hcls, b.cls, t); s.d2h(hinst, b.inst, t); s.d2h(hiw, b.iw, t); s.d2h(hbs, b.bs, t * 3); s.d2h(hnv, b.nv, t);
    for (int o = 0; o < s.n_obj; ++o)
        for (int m = 0; m < hn[o]; ++m) {
            const size_t i = (size_t)o * M + m;
            VotingMaximum vm;
            vm.position = {hpos[i * 3], hpos[i * 3 + 1], hpos[i * 3 + 2]}; vm.weight = hw[i]; vm.classId = (unsigned)hcls[i]; vm.instanceId = (unsigned)hinst[i];
            vm.instanceWeight = hiw[i]; vm.numVotes = hnv[i];
            vm.boundingBox.position = vm.position; vm.boundingBox.size = {hbs[i * 3], hbs[i * 3 + 1], hbs[i * 3 + 2]};
            if (with_quat) vm.boundingBox.rotQuat = {hbq[i * 4], hbq[i * 4 + 1], hbq[i * 4 + 2], hbq[i * 4 + 3]};
            out[o].push_back(vm);
        }
    return true;
}

VotingHough3D::VotingHough3D() {                  // voting_hough_3d.cpp:15-26
    addParameter(m_useInterpolation, "UseInterpolation", true);
    addParameter(m_minCoord, "MinCoord", Vec3d{{-5, -5, -5}});
    addParameter(m_maxCoord, "MaxCoord", Vec3d{{5, 5, 5}});
    addParameter(m_binSize, "BinSize", Vec3d{{0.2, 0.2, 0.2}});
    addParameter(m_relThreshold, "RelThreshold", 0.8f);
}
void VotingHough3D::iFindMaxima(DeviceSession& s, std::vector<std::vector<VotingMaximum>>& out) {
    if (m_single_object_mode) LOG_WARN("SingleObjectMode is not supported with Hough3D - switch to MeanShift to use it!");   // :42-43
    const int C = std::max(1, s.n_classes);
    for (int M : {32, 1024}) {
    ismhip_hough_params P{};
    P.n_classes = C;
    for (int d = 0; d < 3; ++d) { P.min_coord[d] = (float)m_minCoord[d]; P.max_coord[d] = (float)m_maxCoord[d]; }
    // :45-47: MaximaHandler::setRadius(BinSize[0] / 2); the bins become cubes of edge 2 * getSearchDistForClass (= BinSize[0] with "Config")
    P.bin_size = 2.0f * (float)(m_binSize[0] / 2);
    std::vector<float> class_bin = searchDistPerClass((float)(m_binSize[0] / 2), C);
    for (float& b : class_bin) b *= 2.0f;
    P.class_bin_h = class_bin.empty() ? nullptr : class_bin.data();
    P.use_interpolation = m_useInterpolation ? 1 : 0; P.rel_threshold = m_relThreshold;
    P.min_votes_threshold = m_minVotesThreshold; P.min_threshold = m_minThreshold; P.best_k = m_bestK;
    P.max_filter = maxFilter();
    MaximaBuffers B; B.M = M; P.max_maxima = B.M; B.reserve(s.n_obj, C);
    if (m_averageRotation) { P.vote_bbox_quat = s.v_bq.as<float>(); P.max_bbox_quat_out = B.bq.as<float>(); }
    s.check(ismhip_hough3d_maxima(s.ctx, s.n_obj, s.slot_off.data(), s.v_pos.as<float>(), s.v_w.as<float>(), s.v_cls.as<int32_t>(), s.v_inst.as<int32_t>(),
                                  s.v_bs.as<float>(), &P, B.n_max.as<int32_t>(), B.pos.as<float>(), B.w.as<float>(), B.cls.as<int32_t>(), B.inst.as<int32_t>(),
                                  B.iw.as<float>(), B.bs.as<float>(), B.nv.as<int32_t>(), B.score.as<float>()), "ismhip_hough3d_maxima");
    if (collectMaxima(s, B, out, m_averageRotation) || M == 1024) break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Factories (features_factory.h:47-110, keypoints_factory.h:24-38, activation_strategy_factory.h:22-35, voting_factory.h:20-29)
// ---------------------------------------------------------------------------------------------------------------
template <> Features* Factory<Features>::createByType(const std::string& type) {
    if (type == FeaturesSHOT::getTypeStatic()) return new FeaturesSHOT();
    if (type == FeaturesCSHOT::getTypeStatic()) return new FeaturesCSHOT();
    if (type == FeaturesFPFH::getTypeStatic()) return new FeaturesFPFH();
    throw RuntimeException("feature type \"" + type + "\" is outside the MI355X hot path (built: SHOT, CSHOT, FPFH)");
}
template <> Keypoints* Factory<Keypoints>::createByType(const std::string& type) {
    if (type == KeypointsVoxelGrid::getTypeStatic()) return new KeypointsVoxelGrid();
    throw RuntimeException("keypoint type \"" + type + "\" is not built (built: VoxelGrid)");
}
template <> ActivationStrategy* Factory<ActivationStrategy>::createByType(const std::string& type) {
    if (type == ActivationStrategyKNN::getTypeStatic()) return new ActivationStrategyKNN();
    if (type == ActivationStrategyKnnRule::getTypeStatic()) return new ActivationStrategyKnnRule();
    throw RuntimeException("activation strategy \"" + type + "\" is not built (built: KNN, KNNRule)");
}
template <> Voting* Factory<Voting>::createByType(const std::string& type) {
    if (type == VotingMeanShift::getTypeStatic()) return new VotingMeanShift();
    if (type == VotingHough3D::getTypeStatic()) return new VotingHough3D();
    throw RuntimeException("voting type \"" + type + "\" is not built (built: MeanShift, Hough3D)");
}
template <> Codebook* Factory<Codebook>::createByType(const std::string&) { return new Codebook(); }
template <> Clustering* Factory<Clustering>::createByType(const std::string& type) {       // clustering_factory.h
    if (type == ClusteringNone::getTypeStatic()) return new ClusteringNone();
    if (type == ClusteringKMeansCount::getTypeStatic()) return new ClusteringKMeansCount();
    if (type == ClusteringKMeansFactor::getTypeStatic()) return new ClusteringKMeansFactor();
    if (type == ClusteringKMeansThumbRule::getTypeStatic()) return new ClusteringKMeansThumbRule();
    if (type == ClusteringKMeansHartigan::getTypeStatic()) return new ClusteringKMeansHartigan();
    throw RuntimeException("clustering type \"" + type + "\" is not built (built: None, KMeansCount, KMeansFactor, KMeansThumbRule, KMeansHartigan)");
}

// ---------------------------------------------------------------------------------------------------------------
// Clustering (clustering/*.cpp): k-means runs on the device (ismhip_kmeans)
// ---------------------------------------------------------------------------------------------------------------
struct ClusterCenters { DevBuf buf; };
Clustering::Clustering() {}
Clustering::~Clustering() {}
void Clustering::clear() { m_n_centers = 0; m_indices.clear(); m_distances.clear(); }
const float* Clustering::getClusterCentersDevice() const { return m_centers ? m_centers->buf.as<float>() : nullptr; }
void ClusteringNone::process(DeviceSession&, const DeviceFeatures& f, int) {
    m_indices.resize(f.n);
    std::iota(m_indices.begin(), m_indices.end(), 0);
}
ClusteringKMeans::ClusteringKMeans() {             // clustering_kmeans.cpp:18-26
    addParameter(m_iterations, "Iterations", 1000);
    addParameter(m_centersInit, "CentersInit", std::string("FLANN_CENTERS_KMEANSPP"));
    addParameter(m_cbIndex, "CbIndex", 0.5f);      // FLANN's cluster-boundary index: search-time only, without effect on the exact search here
    addParameter(m_seed, "Seed", 0);               // this build's random draws (the reference draws from rand())
}
void ClusteringKMeans::cluster(DeviceSession& s, const DeviceFeatures& f, int metric, int clusterCount) {   // clustering_kmeans.cpp:32-52, .h:53-131
    if (f.n == 0) return;
    if (clusterCount == 0) clusterCount = 1;
    int init;
    if (m_centersInit == "FLANN_CENTERS_RANDOM") init = ISMHIP_CENTERS_RANDOM;
    else if (m_centersInit == "FLANN_CENTERS_GONZALES") init = ISMHIP_CENTERS_GONZALES;
    else if (m_centersInit == "FLANN_CENTERS_KMEANSPP") init = ISMHIP_CENTERS_KMEANSPP;
    else throw BadParamExceptionType<std::string>("invalid flann centers init", m_centersInit);
    if (clusterCount > (int)f.n) {
        LOG_WARN("Desired clusters is higher than available feature count. Creating " << f.n << " individual clusters.");
        clusterCount = (int)f.n;
    }
    LOG_INFO("clustering " << f.n << " features into " << clusterCount << " clusters");
    if (metric != ISMHIP_METRIC_L2SQ)
        LOG_WARN("The k-means algorithm is only defined on euclidean distance. Using other distance metrics may lead to unexpected results.");
    if (!m_centers) m_centers.reset(new ClusterCenters());
    m_centers->buf.reserve((size_t)clusterCount * f.dim * sizeof(float));
    DevBuf assign, dist;
    assign.reserve((size_t)f.n * 4); dist.reserve((size_t)f.n * 4);
    int32_t count = 0, iters = 0;
    s.check(ismhip_kmeans(s.ctx, metric, (int)f.n, f.dim, f.desc.as<float>(), clusterCount, m_iterations, init, (unsigned long long)(long long)m_seed,
                          m_centers->buf.as<float>(), assign.as<int32_t>(), dist.as<float>(), &count, &iters), "ismhip_kmeans");
    if (count != clusterCount) LOG_WARN("Requested " << clusterCount << " but extracted " << count << " clusters instead");
    LOG_INFO("k-means: " << iters << " iterations");
    m_n_centers = count;
    std::vector<int32_t> idx; s.d2h(idx, assign, f.n);
    m_indices.assign(idx.begin(), idx.end());
    s.d2h(m_distances, dist, f.n);
}
ClusteringKMeansCount::ClusteringKMeansCount() { addParameter(m_clusterCount, "ClusterCount", 10); }
void ClusteringKMeansCount::process(DeviceSession& s, const DeviceFeatures& f, int metric) { cluster(s, f, metric, m_clusterCount); }
ClusteringKMeansFactor::ClusteringKMeansFactor() { addParameter(m_clusterFactor, "ClusterFactor", 0.2f); }
void ClusteringKMeansFactor::process(DeviceSession& s, const DeviceFeatures& f, int metric) {
    if (m_clusterFactor > 1) { LOG_WARN("cluster count factor has to be in range [0, 1], setting to 0.5"); m_clusterFactor = 0.5f; }
    cluster(s, f, metric, (int)std::round(f.n * m_clusterFactor));
}
void ClusteringKMeansThumbRule::process(DeviceSession& s, const DeviceFeatures& f, int metric) {
    cluster(s, f, metric, (int)std::round(std::sqrt(f.n / 2.0f)));     // Mardia, Multivariate Analysis, p. 365
}
ClusteringKMeansHartigan::ClusteringKMeansHartigan() { addParameter(m_maxK, "MaxK", 10); }
void ClusteringKMeansHartigan::process(DeviceSession& s, const DeviceFeatures& f, int metric) {   // clustering_kmeans_hartigan.cpp:27-66
    const int maxK = std::max(1, m_maxK);
    std::vector<float> dispersions(maxK);
    for (int i = 0; i < maxK; ++i) {
        cluster(s, f, metric, i + 1);
        float compactness = 0.0f;                   // withinClusterSumOfSquares: distance of every feature to its nearest centre
        for (float d : m_distances) compactness += d;
        dispersions[i] = compactness;
    }
    int bestK = 0; float maxValue = 0;
    for (int i = 0; i + 1 < maxK; ++i) {
        const int numClusters = i + 1;
        const float factor = (float)((int)f.n - numClusters - 1);
        const float index = ((dispersions[i] / dispersions[i + 1]) - 1) * factor;
        if (index > maxValue) { maxValue = index; bestK = i + 1; }
    }
    if (bestK < 1) bestK = 1;                       // the reference would index centers[-1] here
    LOG_INFO("best value for k: " << bestK);
    cluster(s, f, metric, bestK);                   // same seed, same data: the clustering of that pass again
}

// ---------------------------------------------------------------------------------------------------------------
// ImplicitShapeModel (implicit_shape_model.cpp)
// ---------------------------------------------------------------------------------------------------------------
ImplicitShapeModel::ImplicitShapeModel() {        // :91-168 (hot-path subset + the filter switches, which must stay off)
    addParameter(m_distanceType, "DistanceType", std::string("Euclidean"));
    addParameter(m_normal_radius, "NormalRadius", 0.05f);
    addParameter(m_consistent_normals_method, "ConsistentNormalsMethod", 2);
    addParameter(m_num_threads, "NumThreads", 0);
    addParameter(m_bounding_box_type, "BoundingBoxType", std::string("MVBB"));
    addParameter(m_num_kd_trees, "FLANNNumKDTrees", 4);
    addParameter(m_flann_exact_match, "FLANNExactMatch", false);
    addParameter(m_instance_labels_primary, "InstanceLabelsPrimary", true);
    addParameter(m_single_object_mode_legacy, "SingleObjectMode", false);
    addParameter(m_use_smoothing, "UseSmoothing", false);
    addParameter(m_use_sor, "UseStatisticalOutlierRemoval", false);
    addParameter(m_use_ror, "UseRadiusOutlierRemoval", false);
    addParameter(m_use_voxel_filtering, "UseVoxelFiltering", false);
    m_codebook.reset(new Codebook());
    m_keypoints_detector.reset(new KeypointsVoxelGrid());
    m_feature_descriptor.reset(new FeaturesSHOT());
    m_voting.reset(new VotingMeanShift());
}
ImplicitShapeModel::~ImplicitShapeModel() {
    m_codebook.reset();     // releases its device handle while the session is alive
    m_session.reset();
}
void ImplicitShapeModel::iPostInitConfig() {      // :1260-1273
    if (m_distanceType != "Euclidean" && m_distanceType != "ChiSquared") throw BadParamExceptionType<std::string>("invalid distance type", m_distanceType);
    if (!m_flann_exact_match)
        LOG_WARN("FLANNExactMatch is false: the MI355X path always searches exactly (FLANN's randomized kd-forest with 128 checks is not reproduced)");
}
void ImplicitShapeModel::clear() { m_training_clouds.clear(); m_training_instances.clear(); }

Json ImplicitShapeModel::iChildConfigsToJson() const {   // :1070-1083
    Json c = Json::object();
    c["Codebook"] = m_codebook->configToJson();
    c["Keypoints"] = m_keypoints_detector->configToJson();
    c["Features"] = m_feature_descriptor->configToJson();
    if (!m_global_features_cfg.isNull()) c["GlobalFeatures"] = m_global_features_cfg;
    c["Clustering"] = m_clustering ? m_clustering->configToJson() : Json::parse("{\"Type\":\"None\"}");
    c["Voting"] = m_voting->configToJson();
    c["FeatureWeighting"] = m_feature_ranking_cfg.isNull() ? Json::parse("{\"Type\":\"Uniform\"}") : m_feature_ranking_cfg;
    return c;
}
bool ImplicitShapeModel::iChildConfigsFromJson(const Json& c) {   // :1085-1142
    const Json* cb = c.find("Codebook"); const Json* kp = c.find("Keypoints"); const Json* ft = c.find("Features");
    const Json* cl = c.find("Clustering"); const Json* vo = c.find("Voting"); const Json* fw = c.find("FeatureWeighting");
    if (!cb || !kp || !ft || !cl || !vo || !fw) { LOG_ERROR("missing child section (Codebook, Keypoints, Features, Clustering, Voting, FeatureWeighting are required)"); return false; }
    m_codebook.reset(Factory<Codebook>::create(*cb));
    m_keypoints_detector.reset(Factory<Keypoints>::create(*kp));
    m_feature_descriptor.reset(Factory<Features>::create(*ft));
    m_voting.reset(Factory<Voting>::create(*vo));
    m_clustering.reset(Factory<Clustering>::create(*cl)); m_feature_ranking_cfg = *fw;
    if (const Json* gf = c.find("GlobalFeatures")) m_global_features_cfg = *gf;
    const Json* rt = fw->find("Type");
    if (!rt || rt->str != "Uniform") throw RuntimeException("feature ranking type \"" + (rt ? rt->str : std::string()) + "\" is out of scope (built: Uniform)");
    if (m_use_smoothing || m_use_sor || m_use_ror || m_use_voxel_filtering) throw RuntimeException("point cloud pre-filters (smoothing / outlier removal / voxel filtering) are not built");
    return m_codebook && m_keypoints_detector && m_feature_descriptor && m_voting && m_clustering;
}

void ImplicitShapeModel::iSaveData(std::ostream& os) const {      // :1144-1179, as a Boost binary archive (boost_archive.h)
    BoostBinaryOArchive oa(os);
    oa << (unsigned)m_instance_to_class_map.size();
    for (auto& kv : m_instance_to_class_map) oa << kv.first << kv.second;
    m_codebook->save(oa);
    // keypoints, features, global features, clustering: JSONObject::iSaveData writes nothing (json_object.cpp:256-259)
    m_voting->save(oa);
    // feature ranking: nothing
    oa << (unsigned)m_class_labels.size();
    for (auto& kv : m_class_labels) oa << kv.second;
    oa << (unsigned)m_instance_labels.size();
    for (auto& kv : m_instance_labels) oa << kv.second;
}
bool ImplicitShapeModel::iLoadData(std::istream& is) {            // :1181-1237
    BoostBinaryIArchive ia(is);
    if (!ia.ok()) { LOG_ERROR("could not read the data file: " << ia.error()); return false; }
    unsigned size = 0; ia >> size;
    if (!ia.plausible(size, 8)) { LOG_ERROR("could not read the data file: " << ia.error()); return false; }
    m_instance_to_class_map.clear();
    for (unsigned i = 0; i < size; ++i) { unsigned a = 0, b = 0; ia >> a >> b; m_instance_to_class_map.insert({a, b}); }
    if (!m_codebook->load(ia) || !m_voting->load(ia)) { LOG_ERROR("could not load child objects: " << ia.error()); return false; }
    m_n_classes = (int)m_codebook->data().class_sigma.size();
    ia >> size;
    if (!ia.plausible(size, 8)) { LOG_ERROR("could not read the data file: " << ia.error()); return false; }
    m_class_labels.clear();
    for (unsigned i = 0; i < size; ++i) { std::string l; ia >> l; m_class_labels.insert({i, l}); }
    ia >> size;
    if (!ia.plausible(size, 8)) { LOG_ERROR("could not read the data file: " << ia.error()); return false; }
    m_instance_labels.clear();
    for (unsigned i = 0; i < size; ++i) { std::string l; ia >> l; m_instance_labels.insert({i, l}); }
    if (!ia.ok()) { LOG_ERROR("could not read the data file: " << ia.error()); return false; }
    return true;
}

bool ImplicitShapeModel::getDimensions(unsigned class_id, float* out4) const {
    auto it = m_voting->getDimensionsMap().find(class_id);
    auto iv = m_voting->getVarianceMap().find(class_id);
    if (it == m_voting->getDimensionsMap().end() || iv == m_voting->getVarianceMap().end()) return false;
    out4[0] = it->second.first; out4[1] = it->second.second; out4[2] = iv->second.first; out4[3] = iv->second.second;
    return true;
}
void ImplicitShapeModel::setDimensions(unsigned class_id, const float* in4) { m_voting->setDimensions(class_id, in4[0], in4[1], in4[2], in4[3]); }

DeviceSession& ImplicitShapeModel::session() {
    if (!m_session) m_session.reset(new DeviceSession(m_device));
    return *m_session;
}
int ImplicitShapeModel::metric() const { return m_distanceType == "ChiSquared" ? ISMHIP_METRIC_CHI2 : ISMHIP_METRIC_L2SQ; }

bool ImplicitShapeModel::addTrainingModel(const std::string& filename, unsigned class_id, unsigned instance_id) {   // :182-211
    LOG_INFO("adding training model with class id " << class_id << " and instance id " << instance_id);
    std::shared_ptr<PointCloud> c = loadPointCloud(filename);
    if (!c) return false;
    m_training_clouds[class_id].push_back(c); m_training_instances[class_id].push_back(instance_id);
    return true;
}
bool ImplicitShapeModel::addTrainingModel(const PointCloud& cloud, unsigned class_id, unsigned instance_id) {
    m_training_clouds[class_id].push_back(std::make_shared<PointCloud>(cloud)); m_training_instances[class_id].push_back(instance_id);
    return true;
}

static bool firstNormalValid(const PointCloud& c) {   // :615-625 — decided from the FIRST point only
    if (c.nx.size() != c.size() || c.empty()) return false;
    if ((c.nx[0] == 0 && c.ny[0] == 0 && c.nz[0] == 0) || std::isnan(c.nx[0])) return false;
    return true;
}

std::shared_ptr<DeviceFeatures> ImplicitShapeModel::computeFeatures(const std::vector<const PointCloud*>& clouds_in, bool) {   // :733-927
    DeviceSession& s = session();
    auto t0 = std::chrono::steady_clock::now();
    // clouds that come without normals get them on the device (computeNormals :940-1032, ConsistentNormalsMethod 2), then lose the
    // points whose normal is NaN (filterNormals :1034-1075); the features are computed on those completed copies
    std::vector<const PointCloud*> clouds = clouds_in;
    std::vector<std::unique_ptr<PointCloud>> completed;
    const float cell = std::min(m_feature_descriptor->getRadius(), m_feature_descriptor->getType() == "FPFH" ? m_feature_descriptor->getRadius()
                                                                                                           : m_feature_descriptor->getReferenceFrameRadius()) * 0.4f;
    // VoxelGrid keypoints are taken on the device with the batch (ismhip_voxel_keypoints); any other detector, or
    // ISM3D_HOST_KEYPOINTS=1, runs the host implementation per object and uploads its result
    const auto* vg = dynamic_cast<const KeypointsVoxelGrid*>(m_keypoints_detector.get());
    const char* host_kp = getenv("ISM3D_HOST_KEYPOINTS");
    const bool dev_kp = vg && !(host_kp && host_kp[0] == '1');
    auto estimate = [&]() {
        if (m_consistent_normals_method == 2)
            s.check(ismhip_estimate_normals(s.ctx, s.cloud, m_normal_radius, s.nx.as<float>(), s.ny.as<float>(), s.nz.as<float>()), "ismhip_estimate_normals");
        else                                                // :969-1003
            s.check(ismhip_estimate_normals_pca(s.ctx, s.cloud, m_normal_radius, m_consistent_normals_method, s.nx.as<float>(), s.ny.as<float>(), s.nz.as<float>()),
                    "ismhip_estimate_normals_pca");
    };
    {
        std::vector<size_t> need;
        for (size_t i = 0; i < clouds.size(); ++i) if (!firstNormalValid(*clouds[i])) need.push_back(i);
        for (size_t i : need)
            if (clouds[i]->organized) {       // computeNormals :948-966 (pcl::IntegralImageNormalEstimation, AVERAGE_3D_GRADIENT) is not built
                LOG_WARN("organized input cloud without normals: the reference estimates them from the depth image (IntegralImageNormalEstimation); "
                         "here ConsistentNormalsMethod " << m_consistent_normals_method << " is used as for unorganized clouds -- the normals differ");
                break;
            }
        if (!need.empty() && (m_consistent_normals_method < 0 || m_consistent_normals_method > 2))
            throw RuntimeException("input cloud has no normals and ConsistentNormalsMethod " + std::to_string(m_consistent_normals_method) +
                                   " is not built (built: 0 = PCA towards the origin, 1 = PCA away from the centroid, 2 = SHOT reference frames)");
        const char* host_nf = getenv("ISM3D_HOST_NORMAL_FILTER");
        if (need.size() == clouds.size() && dev_kp && !(host_nf && host_nf[0] == '1')) {
            // every cloud of the batch needs normals and the keypoints are taken on the device: the points are uploaded once (with
            // their colours), the normals are estimated, the NaN ones leave (ismhip_filter_normals) and the search surface of the
            // descriptor stage is built over the compacted arrays -- nothing returns to the host in between
            std::vector<std::unique_ptr<PointCloud>> tmp;
            std::vector<const PointCloud*> part;
            for (const PointCloud* src : clouds) {
                tmp.emplace_back(new PointCloud(*src));
                PointCloud& c = *tmp.back();
                c.nx.assign(c.size(), 0.f); c.ny.assign(c.size(), 0.f); c.nz.assign(c.size(), 0.f);
                part.push_back(&c);
            }
            const std::vector<KeypointSet> none(part.size());
            const bool with_color = m_feature_descriptor->needsColor();
            s.uploadBatch(part, &none, m_normal_radius * 0.5f, with_color);
            estimate();
            const size_t n_all = std::max<size_t>(s.pt_off.back(), 1);
            DevBuf* dst[7] = {&s.fx, &s.fy, &s.fz, &s.fnx, &s.fny, &s.fnz, &s.frgba};
            for (int a = 0; a < 7; ++a) if (a < 6 || with_color) dst[a]->reserve(n_all * 4);
            const ismhip_point_arrays in = {s.x.as<float>(), s.y.as<float>(), s.z.as<float>(), s.nx.as<float>(), s.ny.as<float>(), s.nz.as<float>(),
                                            with_color ? s.rgba.as<uint32_t>() : nullptr};
            const ismhip_point_arrays out = {s.fx.as<float>(), s.fy.as<float>(), s.fz.as<float>(), s.fnx.as<float>(), s.fny.as<float>(), s.fnz.as<float>(),
                                             with_color ? s.frgba.as<uint32_t>() : nullptr};
            std::vector<uint32_t> new_off(s.pt_off.size());
            s.check(ismhip_filter_normals(s.ctx, s.n_obj, s.pt_off.data(), &in, &out, new_off.data()), "ismhip_filter_normals");
            s.x.swap(s.fx); s.y.swap(s.fy); s.z.swap(s.fz); s.nx.swap(s.fnx); s.ny.swap(s.fny); s.nz.swap(s.fnz);
            if (with_color) s.rgba.swap(s.frgba);
            s.pt_off = new_off;
            auto t1 = std::chrono::steady_clock::now();
            m_processing_times["normals"] += std::chrono::duration<double, std::milli>(t1 - t0).count();
            s.finishBatch(true, cell, vg->getLeafSize());
            auto t2 = std::chrono::steady_clock::now();
            m_processing_times["keypoints"] += std::chrono::duration<double, std::milli>(t2 - t1).count();
            auto f = (*m_feature_descriptor)(s);
            s.sync();
            m_processing_times["features"] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
            return f;
        }
        if (!need.empty()) {
            // mixed batches and host-side keypoint detectors: the estimated normals come back, the filter runs here and the completed
            // copies are uploaded with the rest
            std::vector<std::unique_ptr<PointCloud>> tmp;
            std::vector<const PointCloud*> part;
            for (size_t i : need) {
                tmp.emplace_back(new PointCloud(*clouds[i]));
                PointCloud& c = *tmp.back();
                c.nx.assign(c.size(), 0.f); c.ny.assign(c.size(), 0.f); c.nz.assign(c.size(), 0.f);
                part.push_back(&c);
            }
            const std::vector<KeypointSet> none(part.size());
            s.uploadBatch(part, &none, m_normal_radius * 0.5f, false);
            estimate();
            std::vector<float> hnx, hny, hnz;
            const size_t n_all = s.pt_off.back();
            s.d2h(hnx, s.nx, n_all); s.d2h(hny, s.ny, n_all); s.d2h(hnz, s.nz, n_all);
            for (size_t k = 0; k < need.size(); ++k) {
                const PointCloud& src = *tmp[k];
                std::unique_ptr<PointCloud> out(new PointCloud());
                const size_t b = s.pt_off[k];
                const bool col = src.rgba.size() == src.size();
                for (size_t i = 0; i < src.size(); ++i) {
                    const float a = hnx[b + i], bb = hny[b + i], cc = hnz[b + i];
                    if (std::isnan(a) || std::isnan(bb) || std::isnan(cc)) continue;
                    out->x.push_back(src.x[i]); out->y.push_back(src.y[i]); out->z.push_back(src.z[i]);
                    out->nx.push_back(a); out->ny.push_back(bb); out->nz.push_back(cc);
                    if (col) out->rgba.push_back(src.rgba[i]);
                }
                clouds[need[k]] = out.get();
                completed.push_back(std::move(out));
            }
        }
    }
    auto t1 = t0;
    if (dev_kp) {
        s.uploadBatch(clouds, nullptr, cell, m_feature_descriptor->needsColor(), vg->getLeafSize());
        t1 = std::chrono::steady_clock::now();
    } else {
        std::vector<KeypointSet> kps;
        for (const PointCloud* c : clouds) kps.push_back((*m_keypoints_detector)(*c));
        t1 = std::chrono::steady_clock::now();
        s.uploadBatch(clouds, &kps, cell, m_feature_descriptor->needsColor());
    }
    m_processing_times["keypoints"] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    auto f = (*m_feature_descriptor)(s);
    s.sync();
    m_processing_times["features"] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    return f;
}

void ImplicitShapeModel::train() {                // :252-500
    if (m_training_clouds.empty()) { LOG_WARN("no training objects found"); return; }
    g_log_info = m_logging;
    if (m_bounding_box_type != "AABB")
        LOG_WARN("BoundingBoxType \"" << m_bounding_box_type << "\": MVBB is not built, using the axis-aligned box centre for the training votes");
    DeviceSession& s = session();
    const int met = metric();
    // all training objects, class by class (std::map order), model by model
    std::vector<const PointCloud*> clouds; std::vector<unsigned> obj_class, obj_inst;
    for (auto& kv : m_training_clouds)
        for (size_t i = 0; i < kv.second.size(); ++i) { clouds.push_back(kv.second[i].get()); obj_class.push_back(kv.first); obj_inst.push_back(m_training_instances[kv.first][i]); }
    m_n_classes = 0;
    for (unsigned c : obj_class) m_n_classes = std::max(m_n_classes, (int)c + 1);
    // features in chunks of objects, gathered on the host as one DeviceFeatures for activation
    auto all = std::make_shared<DeviceFeatures>();
    std::vector<float> hdesc, hlrf, hkx, hky, hkz;
    std::vector<unsigned> fclass, finst, fmodel; std::vector<std::array<float, 3>> fcenter, fbox;
    std::map<unsigned, std::vector<std::array<float, 3>>> box_sizes; std::map<unsigned, std::vector<float>> object_radii;
    const size_t chunk = 32;
    const int D = m_feature_descriptor->getDescriptorLength();
    for (size_t b = 0; b < clouds.size(); b += chunk) {
        std::vector<const PointCloud*> part(clouds.begin() + b, clouds.begin() + std::min(clouds.size(), b + chunk));
        auto f = computeFeatures(part, true);
        std::vector<float> t;
        s.d2h(t, f->desc, (size_t)f->n * D); hdesc.insert(hdesc.end(), t.begin(), t.end());
        s.d2h(t, f->lrf, (size_t)f->n * 9); hlrf.insert(hlrf.end(), t.begin(), t.end());
        s.d2h(t, f->kx, f->n); hkx.insert(hkx.end(), t.begin(), t.end());
        s.d2h(t, f->ky, f->n); hky.insert(hky.end(), t.begin(), t.end());
        s.d2h(t, f->kz, f->n); hkz.insert(hkz.end(), t.begin(), t.end());
        for (size_t o = 0; o < part.size(); ++o) {
            const PointCloud& c = *part[o];
            float mn[3] = {c.x[0], c.y[0], c.z[0]}, mx[3] = {c.x[0], c.y[0], c.z[0]};
            for (size_t i = 0; i < c.size(); ++i) {
                mn[0] = std::min(mn[0], c.x[i]); mx[0] = std::max(mx[0], c.x[i]); mn[1] = std::min(mn[1], c.y[i]); mx[1] = std::max(mx[1], c.y[i]);
                mn[2] = std::min(mn[2], c.z[i]); mx[2] = std::max(mx[2], c.z[i]);
            }
            // Utils::computeAABB (utils.cpp:222-233): size = max - min, position = min + size / 2
            const std::array<float, 3> bsize = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
            const std::array<float, 3> center = {mn[0] + bsize[0] / 2, mn[1] + bsize[1] / 2, mn[2] + bsize[2] / 2};
            // Utils::computeCloudRadius (utils.cpp:302-321): largest distance to the centroid (pcl::compute3DCentroid into a Vector4f)
            float cs[3] = {0, 0, 0};
            for (size_t i = 0; i < c.size(); ++i) { cs[0] += c.x[i]; cs[1] += c.y[i]; cs[2] += c.z[i]; }
            for (int d = 0; d < 3; ++d) cs[d] /= (float)c.size();
            float radius = 0.f;
            for (size_t i = 0; i < c.size(); ++i) {
                const float dx = c.x[i] - cs[0], dy = c.y[i] - cs[1], dz = c.z[i] - cs[2];
                radius = std::max(radius, std::sqrt(dx * dx + dy * dy + dz * dz));
            }
            box_sizes[obj_class[b + o]].push_back(bsize); object_radii[obj_class[b + o]].push_back(radius);
            const uint32_t cnt = f->off[o + 1] - f->off[o];
            fclass.insert(fclass.end(), cnt, obj_class[b + o]); finst.insert(finst.end(), cnt, obj_inst[b + o]);
            fmodel.insert(fmodel.end(), cnt, (unsigned)(b + o)); fcenter.insert(fcenter.end(), cnt, center); fbox.insert(fbox.end(), cnt, bsize);
        }
    }
    all->dim = D; all->n = (uint32_t)fclass.size();
    DeviceSession::h2d(all->desc, hdesc); DeviceSession::h2d(all->lrf, hlrf); DeviceSession::h2d(all->kx, hkx); DeviceSession::h2d(all->ky, hky); DeviceSession::h2d(all->kz, hkz);
    LOG_INFO("activating codewords with " << all->n << " training features");
    m_voting->forwardBoxesAndRadii(box_sizes, object_radii);     // :433: bandwidth hints per class, persisted with the model
    LOG_INFO("clustering");                                      // :445-449
    if (!m_clustering) m_clustering.reset(new ClusteringNone());
    (*m_clustering)(s, *all, met);
    m_codebook->activate(s, *all, fclass, finst, fmodel, fcenter, fbox, met, m_n_classes, *m_clustering);
    LOG_INFO("training done");
}

std::vector<std::vector<VotingMaximum>> ImplicitShapeModel::detectBatch(const std::vector<const PointCloud*>& clouds) {   // detect() :583-712 over a batch
    g_log_info = m_logging;
    if (m_single_object_mode_legacy)
        throw RuntimeException("The parameter for \"single object mode\" must be set inside the \"Voting\" section of the config file. You are using the \"Parameters\" section.");
    auto t_all = std::chrono::steady_clock::now();
    DeviceSession& s = session();
    std::vector<const PointCloud*> nonempty;
    std::vector<int> map;
    for (size_t i = 0; i < clouds.size(); ++i) { if (clouds[i]->empty()) LOG_WARN("point cloud is empty"); else { map.push_back((int)i); nonempty.push_back(clouds[i]); } }
    std::vector<std::vector<VotingMaximum>> out(clouds.size());
    if (nonempty.empty()) return out;
    auto f = computeFeatures(nonempty, false);
    auto t0 = std::chrono::steady_clock::now();
    LOG_INFO("activating codewords and casting votes");
    m_voting->clear();
    m_codebook->castVotes(s, *f, metric(), *m_voting);
    s.sync();
    auto t1 = std::chrono::steady_clock::now();
    m_processing_times["voting"] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    LOG_INFO("finding maxima");
    auto res = m_voting->findMaxima(s);
    auto t2 = std::chrono::steady_clock::now();
    m_processing_times["maxima"] += std::chrono::duration<double, std::milli>(t2 - t1).count();
    m_processing_times["complete"] += std::chrono::duration<double, std::milli>(t2 - t_all).count();
    m_processing_times["normals"] += 0; m_processing_times["flann"] += 0;
    for (size_t i = 0; i < res.size(); ++i) out[map[i]] = res[i];
    return out;
}

std::tuple<std::vector<VotingMaximum>, std::map<std::string, double>> ImplicitShapeModel::detect(const PointCloud& pointCloud, bool hasNormals) {
    if (pointCloud.empty()) { LOG_WARN("point cloud is empty"); return std::make_tuple(std::vector<VotingMaximum>(), m_processing_times); }
    // hasNormals == false (the reference's detect(PointCloud<PointT>) overload, :575-581): whatever normals the cloud carries are ignored and
    // estimated again; hasNormals == true is still checked against the FIRST point (:615-625) inside computeFeatures (firstNormalValid)
    PointCloud stripped;
    const PointCloud* in = &pointCloud;
    if (!hasNormals) { stripped = pointCloud; stripped.nx.assign(stripped.size(), 0.f); stripped.ny.assign(stripped.size(), 0.f); stripped.nz.assign(stripped.size(), 0.f); in = &stripped; }
    auto r = detectBatch({in});
    LOG_INFO("detected " << r[0].size() << " maxima");
    return std::make_tuple(r[0], m_processing_times);
}
bool ImplicitShapeModel::detect(const std::string& filename, std::vector<VotingMaximum>& maxima, std::map<std::string, double>& times) {   // :564-573
    std::shared_ptr<PointCloud> c = loadPointCloud(filename);
    if (!c) return false;
    std::tie(maxima, times) = detect(*c, true);      // "true is assumed because of point type" (:568); the first normal decides (firstNormalValid)
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// file lists (eval_tool/eval_helpers.h:60-177)
// ---------------------------------------------------------------------------------------------------------------
static unsigned convertLabel(const std::string& label, std::map<std::string, unsigned>& labels_map, std::map<unsigned, std::string>& labels_rmap) {
    auto it = labels_map.find(label);
    if (it != labels_map.end()) return it->second;
    const unsigned id = (unsigned)labels_map.size();
    labels_map.insert({label, id}); labels_rmap.insert({id, label});
    return id;
}
FileList parseFileList(const std::string& input_file_name) {
    FileList L;
    std::ifstream infile(input_file_name);
    if (!infile) throw RuntimeException("could not read file list: " + input_file_name);
    std::string file, class_label, instance_label;
    infile >> file; infile >> class_label; infile >> instance_label;
    if (file == "#" && (class_label == "train" || class_label == "test")) {
        L.mode = class_label;
        if (instance_label == "inst") L.using_instances = true;
        if (instance_label == "detection") throw RuntimeException("detection data sets are out of scope (eval_tool_detection)");
    }
    if (L.using_instances) {
        while (infile >> file >> class_label >> instance_label) {
            if (file[0] == '#') continue;
            L.filenames.push_back(file);
            const unsigned c = convertLabel(class_label, L.class_labels_map, L.class_labels_rmap);
            const unsigned i = convertLabel(instance_label, L.instance_labels_map, L.instance_labels_rmap);
            L.instance_to_class_map.insert({i, c});
            L.class_labels.push_back(c); L.instance_labels.push_back(i);
        }
    } else {
        file = instance_label;
        infile >> class_label;
        L.filenames.push_back(file);
        L.class_labels.push_back(convertLabel(class_label, L.class_labels_map, L.class_labels_rmap));
        while (infile >> file >> class_label) {
            if (file[0] == '#') continue;
            L.filenames.push_back(file);
            const unsigned c = convertLabel(class_label, L.class_labels_map, L.class_labels_rmap);
            L.class_labels.push_back(c);
            L.instance_to_class_map.insert({c, c});
        }
        L.instance_labels = L.class_labels;
    }
    return L;
}

}  // namespace ism3d
