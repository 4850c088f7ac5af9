// ism3d.h — C++ host mirror of the reference's plugin interface for the recognition hot path.
//
// Same class names, parameter names, type strings and error behaviour as vseib/point-cloud-donkey
// (paths relative to /root/reference/src/implicit_shape_model):
//   JSONObject / JSONParameter / Factory<T>      utils/json_object.h:31-103, utils/factory.h:20-53
//   Exception hierarchy                          utils/exception.h:21-88
//   Features + SHOT/CSHOT/FPFH                   features/features.h:31-112, features_shot.cpp, features_cshot.cpp, features_fpfh.cpp
//   Keypoints + VoxelGrid                        keypoints/keypoints.h:31-86, keypoints_voxel_grid.cpp:30-46
//   ActivationStrategy(KNN), Codebook            activation_strategy/*.h, codebook/codebook.h:50
//   Voting, VotingMeanShift, Vote, VotingMaximum voting/voting.h:35, voting_mean_shift.cpp, voting_maximum.h:25-88
//   ImplicitShapeModel                           implicit_shape_model.h:82-330
// What differs by design: PCL/Eigen/Boost types are replaced by plain SoA containers, every plugin works on a BATCH of
// objects, and the bodies call the C ABI of libismhip.so (include/ismhip.h) — there is no CPU implementation behind them.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/ismhip.h"
#include "json.h"
#include "boost_archive.h"

namespace ism3d {

// ---- exceptions (utils/exception.h) ----------------------------------------------------------------
class Exception : public std::exception {
public:
    virtual ~Exception() throw() {}
    virtual const char* what() const throw() { return m_message.c_str(); }
protected:
    explicit Exception(std::string m) : m_message(std::move(m)) {}
private:
    std::string m_message;
};
class JSONException : public Exception { public: explicit JSONException(std::string m) : Exception(std::move(m)) {} };
class RuntimeException : public Exception { public: explicit RuntimeException(std::string m) : Exception(std::move(m)) {} };
class BadParamException : public Exception { public: explicit BadParamException(std::string m) : Exception(std::move(m)) {} };
template <typename T>
class BadParamExceptionType : public BadParamException {
public:
    BadParamExceptionType(std::string message, T value) : BadParamException(msg(message, value)) {}
private:
    static std::string msg(std::string m, T v) { std::stringstream s; s << v; if (!s.str().empty()) m += " (Value: " + s.str() + ")"; return m; }
};

// ---- JSON parameters (utils/json_parameter.h, json_parameter_traits.h) -----------------------------
struct JSONParameterBase {
    std::string name;
    virtual ~JSONParameterBase() {}
    virtual void fromJson(const Json* v) = 0;     // nullptr: key missing -> WARN + default (json_parameter_base.cpp:35-45)
    virtual Json toJson() const = 0;
};
template <typename T> struct JSONParameter;
class JSONObject {
public:
    JSONObject();
    virtual ~JSONObject();
    virtual std::string getType() const { return ""; }
    bool writeObject(std::string file);
    bool writeObject(std::string file, std::string fileData);
    bool readObject(std::string file, bool training = false);
    Json configToJson() const;
    bool configFromJson(const Json&);
protected:
    template <typename T> void addParameter(T& param, std::string name, T defaultValue);
    virtual Json iChildConfigsToJson() const { return Json::object(); }
    virtual bool iChildConfigsFromJson(const Json&) { return true; }
    virtual void iSaveData(std::ostream&) const {}
    virtual bool iLoadData(std::istream&) { return true; }
    virtual void iPostInitConfig() {}
    std::string m_output_file_name, m_input_config_file;
private:
    std::vector<JSONParameterBase*> m_params;
};

// typed parameter with the reference's semantics: missing key -> default (+ warning), wrong JSON type -> JSONException
void jsonWarnMissing(const std::string& name);
template <typename T> struct JSONParameterTraits;
template <> struct JSONParameterTraits<bool> {
    static bool ok(const Json& v) { return v.type == Json::Bool; }
    static bool get(const Json& v) { return v.b; }
    static Json put(bool v) { return Json::of(v); }
};
template <> struct JSONParameterTraits<int> {
    static bool ok(const Json& v) { return v.type == Json::Number; }
    static int get(const Json& v) { return (int)v.num; }
    static Json put(int v) { return Json::of(v); }
};
template <> struct JSONParameterTraits<float> {
    static bool ok(const Json& v) { return v.type == Json::Number; }
    static float get(const Json& v) { return (float)v.num; }
    static Json put(float v) { return Json::of((double)v); }
};
template <> struct JSONParameterTraits<double> {
    static bool ok(const Json& v) { return v.type == Json::Number; }
    static double get(const Json& v) { return v.num; }
    static Json put(double v) { return Json::of(v); }
};
template <> struct JSONParameterTraits<std::string> {
    static bool ok(const Json& v) { return v.type == Json::String; }
    static std::string get(const Json& v) { return v.str; }
    static Json put(const std::string& v) { return Json::of(v); }
};
typedef std::array<double, 3> Vec3d;      // Eigen::Vector3d parameters: a JSON array of three numbers (json_parameter_traits.h:107-128)
template <> struct JSONParameterTraits<Vec3d> {
    static bool ok(const Json& v) { return v.type == Json::Array && v.arr.size() == 3 && v.arr[0].type == Json::Number && v.arr[1].type == Json::Number && v.arr[2].type == Json::Number; }
    static Vec3d get(const Json& v) { return Vec3d{{v.arr[0].num, v.arr[1].num, v.arr[2].num}}; }
    static Json put(const Vec3d& v) { Json j = Json::array(); for (double x : v) j.arr.push_back(Json::of(x)); return j; }
};
template <typename T>
struct JSONParameter : JSONParameterBase {
    T& ref; T def;
    JSONParameter(T& r, std::string n, T d) : ref(r), def(d) { name = std::move(n); ref = d; }
    void fromJson(const Json* v) override {
        if (!v) { jsonWarnMissing(name); ref = def; return; }
        if (!JSONParameterTraits<T>::ok(*v)) throw JSONException("invalid type for parameter \"" + name + "\"");
        ref = JSONParameterTraits<T>::get(*v);
    }
    Json toJson() const override { return JSONParameterTraits<T>::put(ref); }
};
template <typename T>
void JSONObject::addParameter(T& param, std::string name, T defaultValue) { m_params.push_back(new JSONParameter<T>(param, std::move(name), defaultValue)); }

// ---- data model -------------------------------------------------------------------------------------
struct PointCloud {          // NaN-free surface with normals (PointXYZRGBNormal split into SoA)
    std::vector<float> x, y, z, nx, ny, nz;
    std::vector<uint32_t> rgba;         // 0x00RRGGBB, empty when the cloud has no colour
    bool organized = false;             // the file was a dense HEIGHT > 1 image: pcl::PointCloud::isOrganized() survives removeNaNFromPointCloud
    size_t size() const { return x.size(); }
    bool empty() const { return x.empty(); }
};
struct KeypointSet { std::vector<float> x, y, z; std::vector<uint32_t> rgba; size_t size() const { return x.size(); } };

struct BoundingBox { std::array<float, 3> position{{0, 0, 0}}; std::array<float, 4> rotQuat{{1, 0, 0, 0}}; std::array<float, 3> size{{0, 0, 0}}; };
struct Vote {                 // voting/voting_maximum.h:25-42
    std::array<float, 3> position; float weight; unsigned classId; unsigned instanceId; int codewordId;
};
struct VotingMaximum {        // voting/voting_maximum.h:51-88
    std::array<float, 3> position{{0, 0, 0}};
    float weight = 0;
    unsigned classId = (unsigned)-1;
    unsigned instanceId = (unsigned)-1;
    float instanceWeight = 0;
    BoundingBox boundingBox;
    int numVotes = 0;
};

class DeviceSession;          // ctx + device buffers (ism3d.cpp)
struct DeviceFeatures;        // device-resident ISMFeature batch: descriptors, LRFs, keypoints, per-object offsets

// ---- Keypoints (keypoints/keypoints.h) -----------------------------------------------------------------
class Keypoints : public JSONObject {
public:
    virtual ~Keypoints() {}
    KeypointSet operator()(const PointCloud& points) const { return iComputeKeypoints(points); }
protected:
    virtual KeypointSet iComputeKeypoints(const PointCloud& points) const = 0;
};
class KeypointsVoxelGrid : public Keypoints {
public:
    KeypointsVoxelGrid();
    static std::string getTypeStatic() { return "VoxelGrid"; }
    std::string getType() const override { return getTypeStatic(); }
    float getLeafSize() const { return m_leafSize; }
protected:
    KeypointSet iComputeKeypoints(const PointCloud& points) const override;
private:
    float m_leafSize;
};

// ---- Features (features/features.h) -------------------------------------------------------------------
class Features : public JSONObject {
public:
    Features();
    virtual ~Features() {}
    // Features::operator() (features.cpp:40-116): LRFs -> drop invalid frames -> iComputeDescriptors -> NaN rows removed.
    // Works on the batch resident in the session; returns the device feature batch.
    std::shared_ptr<DeviceFeatures> operator()(DeviceSession& s) const;
    void setNumThreads(int n) { m_numThreads = n; }
    int getNumThreads() const { return m_numThreads; }
    float getReferenceFrameRadius() const { return m_referenceFrameRadius; }
    virtual float getRadius() const = 0;
    virtual int getDescriptorLength() const = 0;
    virtual bool needsColor() const { return false; }
protected:
    // writes descriptors [nkp x D] for every keypoint of the batch (NaN rows for failures, never an error)
    virtual void iComputeDescriptors(DeviceSession& s, const float* lrf9, float* desc_out, uint32_t* counts_out) const = 0;
    float m_referenceFrameRadius;
    std::string m_referenceFrameType;
    int m_numThreads;
};
#define ISM3D_FEATURE(NAME, TYPESTR, DIM, COLOR)                                                      \
    class NAME : public Features {                                                                    \
    public:                                                                                           \
        NAME();                                                                                       \
        static std::string getTypeStatic() { return TYPESTR; }                                        \
        std::string getType() const override { return getTypeStatic(); }                              \
        float getRadius() const override { return m_radius; }                                         \
        int getDescriptorLength() const override { return DIM; }                                      \
        bool needsColor() const override { return COLOR; }                                            \
    protected:                                                                                        \
        void iComputeDescriptors(DeviceSession& s, const float* lrf9, float* desc_out, uint32_t* counts_out) const override; \
    private:                                                                                          \
        float m_radius;                                                                               \
    };
ISM3D_FEATURE(FeaturesSHOT, "SHOT", 352, false)
ISM3D_FEATURE(FeaturesCSHOT, "CSHOT", 1344, true)
ISM3D_FEATURE(FeaturesFPFH, "FPFH", 33, false)

// ---- activation strategy + codebook ----------------------------------------------------------------------
class ActivationStrategy : public JSONObject {
public:
    ActivationStrategy();
    virtual ~ActivationStrategy() {}
    void setIsDetection() { m_is_detection = true; }
    bool useDistanceRatio() const { return m_use_distance_ratio; }
    float distanceRatioThreshold() const { return m_distance_ratio_threshold; }
    virtual int getK() const = 0;
    // activates every feature of the batch; writes idx/dist [n x columns] on the device and returns the column count
    virtual int activateKNN(DeviceSession& s, const ismhip_codebook* codewords, const DeviceFeatures& f, int metric, int32_t* idx_out, float* dist_out, const float* desc = nullptr) const = 0;
protected:
    bool m_use_distance_ratio; float m_distance_ratio_threshold; bool m_is_detection = false;
};
class ActivationStrategyKNN : public ActivationStrategy {
public:
    ActivationStrategyKNN();
    static std::string getTypeStatic() { return "KNN"; }
    std::string getType() const override { return getTypeStatic(); }
    int getK() const override { return m_k; }
    // activateKNN for a whole feature batch (activation_strategy_knn.h:41-126): idx/dist [n x K] on the device, exact search
    // (FLANNExactMatch semantics). With UseDistanceRatio at detection time and K == 1 the 2-NN ratio test discards matches
    // (idx -1). Returns K.
    int activateKNN(DeviceSession& s, const ismhip_codebook* codewords, const DeviceFeatures& f, int metric, int32_t* idx_out, float* dist_out, const float* desc = nullptr) const override;
private:
    int m_k;
};
class ActivationStrategyKnnRule : public ActivationStrategy {      // activation_strategy/activation_strategy_knn_rule.h:41-152
public:
    ActivationStrategyKnnRule();
    static std::string getTypeStatic() { return "KNNRule"; }
    std::string getType() const override { return getTypeStatic(); }
    int getK() const override { return m_k; }
    // training: plain 1-NN; detection: 3-NN + class-consistency rules. One column.
    int activateKNN(DeviceSession& s, const ismhip_codebook* codewords, const DeviceFeatures& f, int metric, int32_t* idx_out, float* dist_out, const float* desc = nullptr) const override;
private:
    int m_k;
};

// ---- Clustering (clustering/clustering.h:31-99) ----------------------------------------------------------------------------
// operator() clusters the descriptors of all training features; getClusterCenters / getClusterIndices as in the reference, with
// the centres left on the device (they are the rows of the activation codebook). "None": one cluster per feature, no centre matrix.
struct ClusterCenters;         // device matrix [n x dim] (ism3d.cpp)
class Clustering : public JSONObject {
public:
    Clustering();
    virtual ~Clustering();
    void operator()(DeviceSession& s, const DeviceFeatures& f, int metric) { clear(); process(s, f, metric); }
    virtual void clear();
    bool hasCenters() const { return m_n_centers > 0; }
    int getNumCenters() const { return m_n_centers; }
    const float* getClusterCentersDevice() const;
    const std::vector<int>& getClusterIndices() const { return m_indices; }
protected:
    virtual void process(DeviceSession& s, const DeviceFeatures& f, int metric) = 0;
    std::unique_ptr<ClusterCenters> m_centers;
    int m_n_centers = 0;
    std::vector<int> m_indices;
    std::vector<float> m_distances;            // functor distance of every feature to its centre
};
class ClusteringNone : public Clustering {        // clustering_none.cpp:25-35: every feature is its own centre
public:
    static std::string getTypeStatic() { return "None"; }
    std::string getType() const override { return getTypeStatic(); }
protected:
    void process(DeviceSession& s, const DeviceFeatures& f, int metric) override;
};
class ClusteringKMeans : public Clustering {      // clustering_kmeans.{h,cpp}: Iterations, CentersInit, CbIndex (+ Seed: this build's draws)
protected:
    ClusteringKMeans();
    void cluster(DeviceSession& s, const DeviceFeatures& f, int metric, int clusterCount);
    int m_iterations; std::string m_centersInit; float m_cbIndex; int m_seed;
};
class ClusteringKMeansCount : public ClusteringKMeans {         // clustering_kmeans_count.cpp
public:
    ClusteringKMeansCount();
    static std::string getTypeStatic() { return "KMeansCount"; }
    std::string getType() const override { return getTypeStatic(); }
protected:
    void process(DeviceSession& s, const DeviceFeatures& f, int metric) override;
    int m_clusterCount;
};
class ClusteringKMeansFactor : public ClusteringKMeans {        // clustering_kmeans_factor.cpp
public:
    ClusteringKMeansFactor();
    static std::string getTypeStatic() { return "KMeansFactor"; }
    std::string getType() const override { return getTypeStatic(); }
protected:
    void process(DeviceSession& s, const DeviceFeatures& f, int metric) override;
    float m_clusterFactor;
};
class ClusteringKMeansThumbRule : public ClusteringKMeans {     // clustering_kmeans_thumb_rule.cpp
public:
    static std::string getTypeStatic() { return "KMeansThumbRule"; }
    std::string getType() const override { return getTypeStatic(); }
protected:
    void process(DeviceSession& s, const DeviceFeatures& f, int metric) override;
};
class ClusteringKMeansHartigan : public ClusteringKMeans {      // clustering_kmeans_hartigan.cpp
public:
    ClusteringKMeansHartigan();
    static std::string getTypeStatic() { return "KMeansHartigan"; }
    std::string getType() const override { return getTypeStatic(); }
protected:
    void process(DeviceSession& s, const DeviceFeatures& f, int metric) override;
    int m_maxK;
};

struct CodebookData {          // host copy of what Codebook::iSaveData persists (flattened CodewordDistributions)
    int dim = 0;
    std::vector<float> words, word_weight;
    std::vector<uint32_t> vote_offsets{0};
    std::vector<float> vote_xyz, vote_weight, vote_class_weight, vote_bbox_quat, vote_bbox_size;
    std::vector<uint32_t> vote_class, vote_instance;
    std::vector<uint32_t> word_class;      // Codeword::getClassId per word (empty: class of the word's first vote)
    std::vector<int32_t> word_id;          // Codeword::getId per word, ascending (empty: the row index)
    std::vector<int32_t> word_num_features;// Codeword::getNumFeatures (empty: 1)
    std::vector<float> word_keypoint;      // [n_words*3] Codeword::getKeypoint (empty: 0)
    std::vector<float> class_sigma;
    int numWords() const { return dim ? (int)(words.size() / dim) : 0; }
    // consistency of the arrays with each other (a data file is untrusted input); empty string = fine
    std::string validate() const;
};

class Voting;
class Codebook : public JSONObject {
public:
    Codebook();
    ~Codebook();
    std::string getType() const override { return "Codebook"; }
    // training: Codebook::activate (codebook.cpp:64-368); the codewords are the clustering's centres (one per feature for "None")
    void activate(DeviceSession& s, const DeviceFeatures& f, const std::vector<unsigned>& feat_class, const std::vector<unsigned>& feat_instance,
                  const std::vector<unsigned>& feat_model, const std::vector<std::array<float, 3>>& feat_center,
                  const std::vector<std::array<float, 3>>& feat_bbox_size, int metric, int n_classes, const Clustering& clustering);
    // detection: Codebook::castVotes (codebook.cpp:403-555): activate every feature, emit the votes into the voting space
    void castVotes(DeviceSession& s, const DeviceFeatures& f, int metric, Voting& voting) const;
    bool isEmpty() const { return m_data.numWords() == 0; }
    int getSize() const { return m_data.numWords(); }
    int getDim() const { return m_data.dim; }
    const CodebookData& data() const { return m_data; }
    void setData(const CodebookData& d) { m_data = d; m_dirty = true; }
    const ActivationStrategy* getActivationStrategy() const { return m_activationStrategy.get(); }
    void save(BoostBinaryOArchive& oa) const;     // Codebook::iSaveData (codebook.cpp:739-761)
    bool load(BoostBinaryIArchive& ia);           // Codebook::iLoadData (codebook.cpp:763-950)
protected:
    Json iChildConfigsToJson() const override;
    bool iChildConfigsFromJson(const Json&) override;
private:
    void upload(DeviceSession& s) const;
    bool m_useClassWeight, m_useVoteWeight, m_useMatchingWeight, m_useCodewordWeight;
    bool m_use_partial_shot; std::string m_partial_shot_type;
    bool m_use_random_codebook; float m_random_codebook_factor;
    std::unique_ptr<ActivationStrategy> m_activationStrategy;
    CodebookData m_data;
    mutable std::vector<int32_t> m_partial_cols;    // kept descriptor columns when UsePartialShot (empty otherwise)
    mutable bool m_dirty = true;
    mutable ismhip_codebook* m_dev = nullptr;
    mutable DeviceSession* m_dev_session = nullptr;
};

// ---- Voting (voting/voting.h) ----------------------------------------------------------------------------
class Voting : public JSONObject {
public:
    Voting();
    virtual ~Voting() {}
    // Voting::findMaxima for every object of the batch (voting.cpp:79-328)
    std::vector<std::vector<VotingMaximum>> findMaxima(DeviceSession& s);
    void clear();
    bool isSingleObjectMode() const { return m_single_object_mode; }
    // Voting::forwardBoxesAndRadii (voting.cpp:496-551): per class (mean object radius, mean median box edge) and their variances
    void forwardBoxesAndRadii(const std::map<unsigned, std::vector<std::array<float, 3>>>& box_sizes, const std::map<unsigned, std::vector<float>>& object_radii);
    const std::map<unsigned, std::pair<float, float>>& getDimensionsMap() const { return m_dimensions_map; }
    const std::map<unsigned, std::pair<float, float>>& getVarianceMap() const { return m_variance_map; }
    void setDimensions(unsigned class_id, float radius, float box, float radius_var, float box_var) { m_dimensions_map[class_id] = {radius, box}; m_variance_map[class_id] = {radius_var, box_var}; }
    void save(BoostBinaryOArchive& oa) const;     // Voting::iSaveData (voting.cpp:559-614)
    bool load(BoostBinaryIArchive& ia);           // Voting::iLoadData (voting.cpp:616-734)
protected:
    // MaximaHandler::getSearchDistForClass (maxima_handler.cpp:509-521) for every class; empty = the configured radius for all
    std::vector<float> searchDistPerClass(float radius, int n_classes) const;
    std::map<unsigned, std::pair<float, float>> m_dimensions_map, m_variance_map;
    friend class Codebook;
    virtual void iFindMaxima(DeviceSession& s, std::vector<std::vector<VotingMaximum>>& out) = 0;
    // device outputs of ismhip_find_maxima / ismhip_hough3d_maxima -> VotingMaximum lists
    struct MaximaBuffers;                                        // defined in ism3d.cpp (device buffers)
    static bool collectMaxima(DeviceSession& s, MaximaBuffers& b, std::vector<std::vector<VotingMaximum>>& out, bool with_quat);
    int singleObjectMaxType() const;                             // ISMHIP_SOM_* from SingleObjectMode / SingleObjectMaxType
    int maxFilter() const;                                       // ISMHIP_MAXFILTER_* from MaxFilterType
    float m_minThreshold; int m_minVotesThreshold; int m_bestK; bool m_averageRotation;
    std::string m_radiusType; float m_radiusFactor; std::string m_max_filter_type, m_max_type_param;
    bool m_single_object_mode; bool m_use_global_features; bool m_vote_filtering_with_ransac;
};
class VotingHough3D : public Voting {             // voting/voting_hough_3d.{h,cpp}
public:
    VotingHough3D();
    static std::string getTypeStatic() { return "Hough3D"; }
    std::string getType() const override { return getTypeStatic(); }
protected:
    void iFindMaxima(DeviceSession& s, std::vector<std::vector<VotingMaximum>>& out) override;
private:
    bool m_useInterpolation; Vec3d m_minCoord, m_maxCoord, m_binSize; float m_relThreshold;
};
class VotingMeanShift : public Voting {
public:
    VotingMeanShift();
    static std::string getTypeStatic() { return "MeanShift"; }
    std::string getType() const override { return getTypeStatic(); }
protected:
    void iFindMaxima(DeviceSession& s, std::vector<std::vector<VotingMaximum>>& out) override;
private:
    float m_bandwidth, m_threshold; int m_maxIter; std::string m_kernel, m_maxima_suppression_type;
};

// ---- Factory (utils/factory.h) ---------------------------------------------------------------------------
template <typename TClass>
class Factory {
public:
    static TClass* create(const Json& object) {
        if (object.isNull()) return nullptr;
        std::string typeStr;
        if (const Json* t = object.find("Type")) { if (t->type != Json::String) return nullptr; typeStr = t->str; }
        TClass* inst = createByType(typeStr);
        if (!inst || !inst->configFromJson(object)) { delete inst; throw RuntimeException("could not create object of type: \"" + typeStr + "\""); }
        return inst;
    }
private:
    static TClass* createByType(const std::string& type);
};

// ---- ImplicitShapeModel (implicit_shape_model.h:82) -------------------------------------------------------
class ImplicitShapeModel : public JSONObject {
public:
    ImplicitShapeModel();
    ~ImplicitShapeModel();
    std::string getType() const override { return "ImplicitShapeModel"; }

    void clear();
    bool addTrainingModel(const std::string& filename, unsigned class_id, unsigned instance_id);
    bool addTrainingModel(const PointCloud& cloud, unsigned class_id, unsigned instance_id);   // in-memory variant (harness/tests)
    void train();

    std::tuple<std::vector<VotingMaximum>, std::map<std::string, double>> detect(const PointCloud& pointCloud, bool hasNormals = true);
    bool detect(const std::string& filename, std::vector<VotingMaximum>& maxima, std::map<std::string, double>& times);
    // batched fast path (new): descriptors -> kNN -> votes -> maxima stay on the device across many objects
    std::vector<std::vector<VotingMaximum>> detectBatch(const std::vector<const PointCloud*>& clouds);

    const Codebook* getCodebook() const { return m_codebook.get(); }
    const Voting* getVoting() const { return m_voting.get(); }
    void setSignalsState(bool) {}
    void setLogging(bool l) { m_logging = l; }
    void setLabels(std::map<unsigned, std::string>& c, std::map<unsigned, std::string>& i, std::map<unsigned, unsigned>& m) { m_class_labels = c; m_instance_labels = i; m_instance_to_class_map = m; }
    std::map<unsigned, std::string> getClassLabels() { return m_class_labels; }
    std::map<unsigned, std::string> getInstanceLabels() { return m_instance_labels; }
    std::map<unsigned, unsigned> getInstanceClassMap() { return m_instance_to_class_map; }
    bool isInstancePrimaryLabel() { return m_instance_labels_primary; }
    const std::map<std::string, double>& getProcessingTimes() const { return m_processing_times; }
    int numClasses() const { return m_n_classes; }
    // test / tooling access to what travels in the data file
    void setCodebookData(const CodebookData& d, int n_classes) { m_codebook->setData(d); m_n_classes = n_classes; }
    void setLabels(const std::vector<std::string>& cls, const std::vector<std::string>& inst, const std::vector<unsigned>& inst_to_class) {
        m_class_labels.clear(); m_instance_labels.clear(); m_instance_to_class_map.clear();
        for (size_t i = 0; i < cls.size(); ++i) m_class_labels[(unsigned)i] = cls[i];
        for (size_t i = 0; i < inst.size(); ++i) { m_instance_labels[(unsigned)i] = inst[i]; m_instance_to_class_map[(unsigned)i] = inst_to_class[i]; }
    }
    std::string getLabel(int which, unsigned id) const { const auto& m = which == 0 ? m_class_labels : m_instance_labels; auto it = m.find(id); return it == m.end() ? std::string() : it->second; }
    bool getDimensions(unsigned class_id, float* out4) const;
    void setDimensions(unsigned class_id, const float* in4);
    void setDevice(int device) { m_device = device; }

    static std::shared_ptr<PointCloud> loadPointCloud(const std::string& file);

protected:
    Json iChildConfigsToJson() const override;
    bool iChildConfigsFromJson(const Json&) override;
    void iSaveData(std::ostream&) const override;
    bool iLoadData(std::istream&) override;
    void iPostInitConfig() override;

private:
    DeviceSession& session();
    int metric() const;
    std::shared_ptr<DeviceFeatures> computeFeatures(const std::vector<const PointCloud*>& clouds, bool is_training);

    std::string m_distanceType, m_bounding_box_type;
    float m_normal_radius; int m_consistent_normals_method, m_num_threads, m_num_kd_trees;
    bool m_flann_exact_match, m_instance_labels_primary, m_single_object_mode_legacy;
    bool m_use_smoothing, m_use_sor, m_use_ror, m_use_voxel_filtering;
    std::unique_ptr<Codebook> m_codebook;
    std::unique_ptr<Keypoints> m_keypoints_detector;
    std::unique_ptr<Features> m_feature_descriptor;
    std::unique_ptr<Voting> m_voting;
    std::unique_ptr<Clustering> m_clustering;
    Json m_feature_ranking_cfg, m_global_features_cfg;
    std::map<unsigned, std::vector<std::shared_ptr<PointCloud>>> m_training_clouds;     // class -> models
    std::map<unsigned, std::vector<unsigned>> m_training_instances;
    std::map<unsigned, std::string> m_class_labels, m_instance_labels;
    std::map<unsigned, unsigned> m_instance_to_class_map;
    std::map<std::string, double> m_processing_times;
    int m_n_classes = 0;
    bool m_logging = true;
    int m_device = 0;
    std::unique_ptr<DeviceSession> m_session;
};

// list files of eval_tool (eval_tool/eval_helpers.h:100-177)
struct FileList {
    std::vector<std::string> filenames; std::vector<unsigned> class_labels, instance_labels; std::string mode; bool using_instances = false;
    std::map<std::string, unsigned> class_labels_map, instance_labels_map; std::map<unsigned, std::string> class_labels_rmap, instance_labels_rmap;
    std::map<unsigned, unsigned> instance_to_class_map;
};
FileList parseFileList(const std::string& input_file_name);

}  // namespace ism3d
