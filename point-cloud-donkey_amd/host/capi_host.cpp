// capi_host.cpp — small C shim over the C++ host classes so that the Python test-suite can drive ImplicitShapeModel
// (train / write / read / detectBatch) through ctypes. Not part of the drop-in boundary (that is include/ismhip.h).
#include <cstring>
#include <string>

#include "ism3d.h"

using namespace ism3d;

static thread_local std::string g_err;
static PointCloud makeCloud(int n, const float* x, const float* y, const float* z, const float* nx, const float* ny, const float* nz, const uint32_t* rgba) {
    PointCloud c;
    c.x.assign(x, x + n); c.y.assign(y, y + n); c.z.assign(z, z + n); c.nx.assign(nx, nx + n); c.ny.assign(ny, ny + n); c.nz.assign(nz, nz + n);
    if (rgba) c.rgba.assign(rgba, rgba + n);
    return c;
}
#define GUARD(...) try { __VA_ARGS__ } catch (const std::exception& e) { g_err = e.what(); return -1; }

extern "C" {
const char* ism3d_last_error() { return g_err.c_str(); }
void* ism3d_new() { return new ImplicitShapeModel(); }
void ism3d_delete(void* m) { delete (ImplicitShapeModel*)m; }
int ism3d_set_logging(void* m, int on) { ((ImplicitShapeModel*)m)->setLogging(on != 0); return 0; }
int ism3d_read(void* m, const char* file, int training) { GUARD(return ((ImplicitShapeModel*)m)->readObject(file, training != 0) ? 0 : -2;) }
int ism3d_write(void* m, const char* file) { GUARD(return ((ImplicitShapeModel*)m)->writeObject(file) ? 0 : -2;) }
int ism3d_config_from_json(void* m, const char* text) { GUARD(Json j = Json::parse(text); return ((ImplicitShapeModel*)m)->configFromJson(j) ? 0 : -2;) }
int ism3d_config_to_json(void* m, char* out, int cap) {
    GUARD(std::string s = ((ImplicitShapeModel*)m)->configToJson().dump(1); if ((int)s.size() + 1 > cap) return (int)s.size() + 1; std::memcpy(out, s.c_str(), s.size() + 1); return 0;)
}
int ism3d_add_training(void* m, int n, const float* x, const float* y, const float* z, const float* nx, const float* ny, const float* nz,
                       const uint32_t* rgba, unsigned class_id, unsigned instance_id) {
    GUARD(return ((ImplicitShapeModel*)m)->addTrainingModel(makeCloud(n, x, y, z, nx, ny, nz, rgba), class_id, instance_id) ? 0 : -2;)
}
int ism3d_add_training_file(void* m, const char* file, unsigned class_id, unsigned instance_id) {
    GUARD(return ((ImplicitShapeModel*)m)->addTrainingModel(std::string(file), class_id, instance_id) ? 0 : -2;)
}
// loads a cloud file the way addTrainingModel does and copies it out (reader tests); returns the point count, -2 on failure.
// Arrays may be NULL to query the size; rgba_out receives 0 for clouds without colour.
int ism3d_load_cloud(const char* file, int cap, float* x, float* y, float* z, float* nx, float* ny, float* nz, uint32_t* rgba_out) {
    GUARD(
        auto c = ImplicitShapeModel::loadPointCloud(std::string(file));
        if (!c) return -2;
        const int n = (int)c->size();
        for (int i = 0; i < n && i < cap; ++i) {
            if (x) { x[i] = c->x[i]; y[i] = c->y[i]; z[i] = c->z[i]; }
            if (nx) { nx[i] = c->nx[i]; ny[i] = c->ny[i]; nz[i] = c->nz[i]; }
            if (rgba_out) rgba_out[i] = c->rgba.size() == c->size() ? c->rgba[i] : 0u;
        }
        return n;)
}
int ism3d_train(void* m) { GUARD(((ImplicitShapeModel*)m)->train(); return 0;) }
int ism3d_codebook_size(void* m) { return ((ImplicitShapeModel*)m)->getCodebook()->getSize(); }
int ism3d_num_classes(void* m) { return ((ImplicitShapeModel*)m)->numClasses(); }
// copies the codebook tables out (for cross-checks against the Python harness); pass NULL to query sizes
int ism3d_codebook_get(void* m, float* words, float* vote_xyz, uint32_t* vote_class, float* class_sigma) {
    const CodebookData& d = ((ImplicitShapeModel*)m)->getCodebook()->data();
    if (words) std::memcpy(words, d.words.data(), d.words.size() * 4);
    if (vote_xyz) std::memcpy(vote_xyz, d.vote_xyz.data(), d.vote_xyz.size() * 4);
    if (vote_class) std::memcpy(vote_class, d.vote_class.data(), d.vote_class.size() * 4);
    if (class_sigma) std::memcpy(class_sigma, d.class_sigma.data(), d.class_sigma.size() * 4);
    return (int)d.vote_class.size();
}
// installs a codebook from flat arrays (persistence tests without a GPU); optional arrays may be NULL
int ism3d_codebook_set(void* m, int n_words, int dim, const float* words, const int32_t* word_id, const uint32_t* word_class, const float* word_weight,
                       const float* word_keypoint, const uint32_t* vote_off, const float* vote_xyz, const float* vote_weight, const float* vote_class_weight,
                       const uint32_t* vote_class, const uint32_t* vote_instance, const float* vote_bbox_quat, const float* vote_bbox_size,
                       int n_classes, const float* class_sigma) {
    GUARD(
        CodebookData d; d.dim = dim;
        const size_t nv = vote_off[n_words];
        d.words.assign(words, words + (size_t)n_words * dim);
        if (word_id) d.word_id.assign(word_id, word_id + n_words);
        if (word_class) d.word_class.assign(word_class, word_class + n_words);
        if (word_weight) d.word_weight.assign(word_weight, word_weight + n_words);
        if (word_keypoint) d.word_keypoint.assign(word_keypoint, word_keypoint + (size_t)n_words * 3);
        d.vote_offsets.assign(vote_off, vote_off + n_words + 1);
        d.vote_xyz.assign(vote_xyz, vote_xyz + nv * 3);
        if (vote_weight) d.vote_weight.assign(vote_weight, vote_weight + nv);
        if (vote_class_weight) d.vote_class_weight.assign(vote_class_weight, vote_class_weight + nv);
        d.vote_class.assign(vote_class, vote_class + nv); d.vote_instance.assign(vote_instance, vote_instance + nv);
        if (vote_bbox_quat) d.vote_bbox_quat.assign(vote_bbox_quat, vote_bbox_quat + nv * 4);
        if (vote_bbox_size) d.vote_bbox_size.assign(vote_bbox_size, vote_bbox_size + nv * 3);
        d.class_sigma.assign(class_sigma, class_sigma + n_classes);
        const std::string why = d.validate();
        if (!why.empty()) { g_err = why; return -2; }
        ((ImplicitShapeModel*)m)->setCodebookData(d, n_classes);
        return 0;)
}
// every persisted table of the codebook; arrays may be NULL. Returns the number of votes; *n_words_out / *dim_out the shape.
int ism3d_codebook_get_all(void* m, int* n_words_out, int* dim_out, int* n_classes_out, float* words, int32_t* word_id, uint32_t* word_class, float* word_weight,
                           float* word_keypoint, uint32_t* vote_off, float* vote_xyz, float* vote_weight, float* vote_class_weight, uint32_t* vote_class,
                           uint32_t* vote_instance, float* vote_bbox_quat, float* vote_bbox_size, float* class_sigma) {
    const CodebookData& d = ((ImplicitShapeModel*)m)->getCodebook()->data();
    const int nw = d.numWords();
    if (n_words_out) *n_words_out = nw;
    if (dim_out) *dim_out = d.dim;
    if (n_classes_out) *n_classes_out = (int)d.class_sigma.size();
    auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    cp(words, d.words); cp(word_id, d.word_id); cp(word_class, d.word_class); cp(word_weight, d.word_weight); cp(word_keypoint, d.word_keypoint);
    cp(vote_off, d.vote_offsets); cp(vote_xyz, d.vote_xyz); cp(vote_weight, d.vote_weight); cp(vote_class_weight, d.vote_class_weight);
    cp(vote_class, d.vote_class); cp(vote_instance, d.vote_instance); cp(vote_bbox_quat, d.vote_bbox_quat); cp(vote_bbox_size, d.vote_bbox_size);
    cp(class_sigma, d.class_sigma);
    return (int)d.vote_class.size();
}
// label maps and the per-class size hints (Voting::forwardBoxesAndRadii) that travel with the model
int ism3d_set_labels(void* m, int n_classes, const char* const* class_labels, int n_inst, const char* const* inst_labels, const unsigned* inst_to_class) {
    GUARD(((ImplicitShapeModel*)m)->setLabels(std::vector<std::string>(class_labels, class_labels + n_classes), std::vector<std::string>(inst_labels, inst_labels + n_inst),
                                             std::vector<unsigned>(inst_to_class, inst_to_class + n_inst)); return 0;)
}
int ism3d_get_label(void* m, int which, unsigned id, char* out, int cap) {
    GUARD(std::string s = ((ImplicitShapeModel*)m)->getLabel(which, id); if ((int)s.size() + 1 > cap) return -2; std::memcpy(out, s.c_str(), s.size() + 1); return (int)s.size();)
}
int ism3d_dimensions(void* m, unsigned class_id, float* out4) {     // object radius, median box edge, and their variances; -2 when absent
    GUARD(return ((ImplicitShapeModel*)m)->getDimensions(class_id, out4) ? 0 : -2;)
}
int ism3d_set_dimensions(void* m, unsigned class_id, const float* in4) { GUARD(((ImplicitShapeModel*)m)->setDimensions(class_id, in4); return 0;) }

// detectBatch over concatenated SoA arrays; outputs per object up to max_maxima records sorted by weight
int ism3d_detect_batch(void* m, int n_obj, const uint32_t* pt_off, const float* x, const float* y, const float* z, const float* nx, const float* ny,
                       const float* nz, const uint32_t* rgba, int max_maxima, int32_t* n_out, float* pos_out, float* weight_out, int32_t* cls_out,
                       int32_t* inst_out, int32_t* nvotes_out, float* quat_out /* [n_obj*max_maxima*4] boundingBox.rotQuat, may be NULL */,
                       int32_t* n_total_out /* [n_obj] maxima the model returned (may exceed max_maxima), may be NULL */) {
    GUARD(
        std::vector<PointCloud> clouds(n_obj);
        std::vector<const PointCloud*> ptrs;
        for (int o = 0; o < n_obj; ++o) {
            const uint32_t s = pt_off[o]; const int n = (int)(pt_off[o + 1] - s);
            clouds[o] = makeCloud(n, x + s, y + s, z + s, nx + s, ny + s, nz + s, rgba ? rgba + s : nullptr);
            ptrs.push_back(&clouds[o]);
        }
        auto res = ((ImplicitShapeModel*)m)->detectBatch(ptrs);
        for (int o = 0; o < n_obj; ++o) {
            const int nm = std::min((int)res[o].size(), max_maxima);
            n_out[o] = nm;
            if (n_total_out) n_total_out[o] = (int)res[o].size();
            for (int i = 0; i < max_maxima; ++i) {
                const size_t t = (size_t)o * max_maxima + i;
                const bool ok = i < nm;
                pos_out[t * 3] = ok ? res[o][i].position[0] : 0; pos_out[t * 3 + 1] = ok ? res[o][i].position[1] : 0; pos_out[t * 3 + 2] = ok ? res[o][i].position[2] : 0;
                weight_out[t] = ok ? res[o][i].weight : 0; cls_out[t] = ok ? (int)res[o][i].classId : -1; inst_out[t] = ok ? (int)res[o][i].instanceId : -1;
                nvotes_out[t] = ok ? res[o][i].numVotes : 0;
                if (quat_out) for (int d = 0; d < 4; ++d) quat_out[t * 4 + d] = ok ? res[o][i].boundingBox.rotQuat[d] : (d == 0 ? 1.f : 0.f);
            }
        }
        return 0;)
}
int ism3d_detect_file(void* m, const char* file, int32_t* cls_out, float* weight_out) {
    GUARD(std::vector<VotingMaximum> maxima; std::map<std::string, double> times;
          if (!((ImplicitShapeModel*)m)->detect(std::string(file), maxima, times)) return -2;
          *cls_out = maxima.empty() ? -1 : (int)maxima[0].classId; *weight_out = maxima.empty() ? 0.f : maxima[0].weight; return 0;)
}
}
