/*
 * ism_oracle.h — CPU restatement (oracle) of the implicit_shape_model recognition hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / timed CPU baseline. The product path (libismhip.so) never links or calls it.
 *
 * PARITY UNPINNED by the reference: vseib/point-cloud-donkey ships no tests, golden vectors or
 * fixtures and its hot loops live in PCL 1.10 / FLANN 1.9.1 / Eigen 3.3, none of
 * which is present here, so the reference cannot be built (oracle/README.md). The arithmetic below
 * restates the published PCL/FLANN algorithms in the reference's own operation order and is pinned
 * by hand-derived known-answer vectors (tests/golden/, generator committed).
 *
 * All pointers are HOST pointers; signatures mirror include/ismhip.h with the ctx / handles removed.
 */
#ifndef ISM_ORACLE_H_
#define ISM_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ismref_maxima_params {
    int   n_classes;
    const float* class_bandwidth; /* may be NULL */
    float bandwidth;
    float threshold;
    int   max_iter;
    int   kernel;        /* 0 gaussian, 1 uniform */
    int   suppression;   /* 0 average, 1 suppress, 2 none */
    int   min_votes_threshold;
    float min_threshold;
    int   best_k;
    int   max_maxima;
    int   max_filter;    /* 0 none, 1 "Simple", 2 "Merge" (MaximaHandler::filterMaxima, maxima_handler.cpp:272-440) */
    const float* vote_bbox_quat;    /* [n_slots*4] or NULL: Voting.AverageRotation (voting.cpp:210-215) */
    float* max_bbox_quat_out;       /* [n_obj*max_maxima*4] or NULL */
    int   single_object_max_type;   /* 0 mean shift, 1 BANDWIDTH, 2 MODEL_RADIUS, 3 COMPLETE_VOTING_SPACE (voting_mean_shift.cpp:124-157) */
    const float* object_centroid;   /* [n_obj*3] */
    const float* object_radius;     /* [n_obj] */
} ismref_maxima_params;

void ismref_set_num_threads(int n);
int  ismref_get_num_threads(void);

/* radius search used by every descriptor (pcl::search::KdTree::radiusSearch, sorted): returns the
 * number of neighbours with d^2 < r^2; writes up to cap indices (into the object's points) and squared
 * distances in ascending (d^2, index) order. */
int  ismref_radius_search(int n, const float* x, const float* y, const float* z,
                          float qx, float qy, float qz, float radius, int cap, int32_t* idx_out, float* d2_out);

int  ismref_shot_lrf(int n_obj, const uint32_t* pt_offsets, const float* x, const float* y, const float* z,
                     const uint32_t* kp_offsets, const float* kpx, const float* kpy, const float* kpz,
                     float radius, float* lrf9_out);
int  ismref_shot352(int n_obj, const uint32_t* pt_offsets, const float* x, const float* y, const float* z,
                    const float* nx, const float* ny, const float* nz,
                    const uint32_t* kp_offsets, const float* kpx, const float* kpy, const float* kpz,
                    const float* lrf9, float radius, float* desc_out, uint32_t* neighbour_count_out);
int  ismref_cshot1344(int n_obj, const uint32_t* pt_offsets, const float* x, const float* y, const float* z,
                      const float* nx, const float* ny, const float* nz, const uint32_t* rgba,
                      const uint32_t* kp_offsets, const float* kpx, const float* kpy, const float* kpz,
                      const uint32_t* kp_rgba, const float* lrf9, float radius, float* desc_out,
                      uint32_t* neighbour_count_out);
int  ismref_fpfh33(int n_obj, const uint32_t* pt_offsets, const float* x, const float* y, const float* z,
                   const float* nx, const float* ny, const float* nz,
                   const uint32_t* kp_offsets, const float* kpx, const float* kpy, const float* kpz,
                   float radius, float* desc_out, uint32_t* neighbour_count_out);
int  ismref_centroids(int n_obj, const uint32_t* pt_offsets, const float* x, const float* y, const float* z,
                      float* centroid_out);
int  ismref_center_dist(int n_obj, const uint32_t* pt_offsets, const float* x, const float* y, const float* z,
                        const uint32_t* kp_offsets, const float* kpx, const float* kpy, const float* kpz, float* out);
void ismref_rgb2lab(uint32_t rgba, float* L, float* a, float* b);
/* single pair feature of FPFH (pcl::computePairFeatures); returns 1 when valid */
int  ismref_pair_features(const float* p1, const float* n1, const float* p2, const float* n2, float* f4_out);

float ismref_distance(int metric, int dim, const float* a, const float* b);
int  ismref_knn(int metric, int n_words, int dim, const float* words, int nq, const float* q, int k,
                int32_t* idx_out, float* dist_out);
int  ismref_knn_ratio(int metric, int n_words, int dim, const float* words, int nq, const float* q,
                      float ratio_threshold, int32_t* idx_out, float* dist_out);

/* ActivationStrategyKnnRule::activateKNN, detection branch (activation_strategy/activation_strategy_knn_rule.h:79-118) */
int  ismref_knn_rule(int metric, int n_words, int dim, const float* words, const uint32_t* word_class, int nq, const float* q,
                     float ratio_threshold, int32_t* idx_out, float* dist_out);

/* Utils::rotateInto / rotateBack / getRotQuaternion (utils/utils.cpp:136-178) */
void ismref_rot_quaternion(const float* lrf9, float* quat_wxyz_out);
void ismref_rotate_into(const float* lrf9, const float* v, float* out);
void ismref_rotate_back(const float* lrf9, const float* v, float* out);

int  ismref_cast_votes(int n_words, int dim, const float* word_weight,
                       const uint32_t* vote_offsets, const float* vote_xyz, const float* vote_weight,
                       const float* vote_class_weight, const uint32_t* vote_class, const uint32_t* vote_instance,
                       const float* vote_bbox_quat, const float* vote_bbox_size,
                       int n_classes, const float* class_sigma, uint32_t weight_flags,
                       int nq, const float* lrf9, const float* kpx, const float* kpy, const float* kpz,
                       int k, const int32_t* idx, const float* dist,
                       float* vote_pos_out, float* vote_weight_out, int32_t* vote_class_out,
                       int32_t* vote_instance_out, int32_t* vote_codeword_out,
                       float* vote_bbox_quat_out, float* vote_bbox_size_out);

int  ismref_find_maxima(int n_obj, const uint32_t* slot_offsets,
                        const float* vote_pos, const float* vote_weight, const int32_t* vote_class,
                        const int32_t* vote_instance, const float* vote_bbox_size,
                        const ismref_maxima_params* params,
                        int32_t* n_maxima_out, float* max_pos_out, float* max_weight_out,
                        int32_t* max_class_out, int32_t* max_instance_out, float* max_instance_weight_out,
                        float* max_bbox_size_out, int32_t* max_n_votes_out, float* class_score_out);

/* ConsistentNormalsMethod 0 / 1: PCA normals of pcl::NormalEstimationOMPWithEigVals (implicit_shape_model.cpp:969-1011); original point order */
int  ismref_pca_normals(int n_obj, const uint32_t* pt_offsets, const float* x, const float* y, const float* z, float radius, int orientation,
                        float* nx_out, float* ny_out, float* nz_out);

/* VotingHough3D (voting/voting_hough_3d.cpp:33-95) over pcl::recognition::HoughSpace3D; same outputs as
 * ismref_find_maxima. Layout identical to ismhip_hough_params. */
typedef struct ismref_hough_params {
    int   n_classes;
    float min_coord[3];
    float max_coord[3];
    float bin_size;
    const float* class_bin;          /* [n_classes] or NULL */
    int   use_interpolation;
    float rel_threshold;
    int   min_votes_threshold;
    float min_threshold;
    int   best_k;
    int   max_maxima;
    int   max_filter;
    const float* vote_bbox_quat;
    float* max_bbox_quat_out;
} ismref_hough_params;
int  ismref_hough3d_maxima(int n_obj, const uint32_t* slot_offsets,
                           const float* vote_pos, const float* vote_weight, const int32_t* vote_class,
                           const int32_t* vote_instance, const float* vote_bbox_size,
                           const ismref_hough_params* params,
                           int32_t* n_maxima_out, float* max_pos_out, float* max_weight_out,
                           int32_t* max_class_out, int32_t* max_instance_out, float* max_instance_weight_out,
                           float* max_bbox_size_out, int32_t* max_n_votes_out, float* class_score_out);

/* mean-shift building blocks exposed for known-answer tests (voting_mean_shift.cpp:431-481, 331-376) */
int  ismref_create_seeds(int n, const float* pos, const float* w, float bin_size, int cap,
                         float* seed_pos_out, float* seed_w_out);

/* pcl::VoxelGrid centroids (keypoints/keypoints_voxel_grid.cpp:30-46): returns number of keypoints */
int  ismref_voxel_grid(int n, const float* x, const float* y, const float* z, const uint32_t* rgba,
                       float leaf, int cap, float* kx, float* ky, float* kz, uint32_t* krgba);

/* training-side helper used to build synthetic codebooks the way the reference's train() does with
 * Clustering "None" and KNN K=1 (implicit_shape_model.cpp:437-490, codebook.cpp:64-224):
 * class sigma = sample variance of distances between the first <=sqrt(n) features of the class and the
 * first <=sqrt(n) activated codewords (codebook.cpp:94-193). feat_class[n] ascending by class is NOT required. */
int  ismref_class_sigmas(int metric, int dim, int n_feat, const float* feats, const uint32_t* feat_class,
                         const uint32_t* feat_model, const int32_t* activated_word, int n_words, const float* words,
                         int n_classes, float* sigma_out);

/* Codebook::activate (codebook/codebook.cpp:64-368) for one codeword per training feature: exact kNN activation, class sigma^2,
 * K = 1 clean-up, vote CSR, CodewordDistribution::computeWeights, statistical class weights (term1 * term2 * term3).
 * Capacities (m = number of codewords): word_src[m], vote_off[m+1], vote_feature / vote_weight / vote_class_weight [n*k], vote_xyz[n*k*3]. */
int  ismref_activate(int metric, int dim, int n, const float* feats, const float* lrf9, const float* kx, const float* ky, const float* kz,
                     const uint32_t* feat_class, const uint32_t* feat_model, const float* feat_center,
                     int n_codewords, const float* codewords /* NULL: the features themselves */,
                     int k, int clean_up, int n_classes,
                     int32_t* n_words_out, uint32_t* word_src, uint32_t* vote_off, uint32_t* vote_feature, float* vote_xyz,
                     float* vote_weight, float* vote_class_weight, float* class_sigma);

/* ClusteringKMeans::cluster (clustering/clustering_kmeans.h:53-131): one level of FLANN's k-means + nearest centre per feature;
 * centers_init 0 RANDOM, 1 GONZALES, 2 KMEANSPP. FLANN is EXTERNAL and random: the draws and integer sums are the build's own (see .cpp). */
int  ismref_kmeans(int metric, int n, int dim, const float* feats, int n_clusters, int max_iterations, int centers_init, uint64_t seed,
                   float* centers_out, int32_t* assign_out, float* dist_out, int32_t* n_clusters_out, int32_t* iterations_out);

#ifdef __cplusplus
}
#endif
#endif
