/*
 * ismhip.h — C ABI of the MI355X-native implicit_shape_model recognition hot path.
 *
 * This is the drop-in boundary: every entry point replaces one plugin seam of the
 * reference (vseib/point-cloud-donkey, paths relative to /root/reference/src/implicit_shape_model).
 * The reference has no FFI of its own (its seams are C++ virtuals created by Factory<T>,
 * utils/factory.h:24-46), so the functions below are what a maintainer binds from the
 * plugin classes; INTEGRATION.md shows the stubs.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, int status (0 = ok, <0 = error); no exception crosses.
 *   - pointers are DEVICE pointers unless the parameter name ends in _h (host).
 *   - every call takes an ismhip_ctx (device, stream, scratch arena) and is asynchronous on the
 *     ctx stream; outputs are valid after ismhip_sync() or after a stream-ordered consumer.
 *   - one ctx per host thread / GPU; calls on different ctxs are re-entrant.
 *   - object batches: "n_obj" objects are concatenated; X_offsets_h[n_obj+1] gives the element
 *     range of each object in the concatenated arrays (points, keypoints/features, vote slots).
 *   - there is exactly ONE back end (HIP, gfx950). If no GPU is present ismhip_ctx_create fails.
 */
#ifndef ISMHIP_H_
#define ISMHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISMHIP_ABI_VERSION 4

#define ISMHIP_OK               0
#define ISMHIP_ERR_INVALID     -1   /* bad argument (null pointer, negative size, unsupported value) */
#define ISMHIP_ERR_HIP         -2   /* a HIP runtime call failed; see ismhip_last_error */
#define ISMHIP_ERR_NOMEM       -3   /* device allocation failed */
#define ISMHIP_ERR_UNSUPPORTED -4   /* valid in the reference, not built here */
#define ISMHIP_ERR_NODEVICE    -5   /* no gfx950 device visible: the product path has no CPU fallback */

/* distance metric: utils/distance.h:45,65 (FLANN functors; L2 is SQUARED, no sqrt) */
#define ISMHIP_METRIC_L2SQ 0
#define ISMHIP_METRIC_CHI2 1

/* vote weight flags: codebook/codebook.cpp:32-35 */
#define ISMHIP_W_CLASS    1u
#define ISMHIP_W_VOTE     2u
#define ISMHIP_W_MATCHING 4u
#define ISMHIP_W_CODEWORD 8u

/* mean-shift kernel: voting/voting_mean_shift.cpp:378-417 */
#define ISMHIP_KERNEL_GAUSSIAN 0
#define ISMHIP_KERNEL_UNIFORM  1
/* maxima suppression: voting/voting_mean_shift.cpp:99-122 */
#define ISMHIP_SUPPRESS_AVERAGE  0
#define ISMHIP_SUPPRESS_SUPPRESS 1
#define ISMHIP_SUPPRESS_NONE     2
/* inter-class maxima filter: voting/maxima_handler.cpp:272-296 ("Simple" = greedy non-maximum suppression over ALL classes inside
 * the search radius, suppressNeighborMaxima2 :227-268; "Merge" = mergeAndFilterMaxima :298-..., merging the maxima of the SAME class inside the radius first) */
#define ISMHIP_MAXFILTER_NONE   0
#define ISMHIP_MAXFILTER_SIMPLE 1
#define ISMHIP_MAXFILTER_MERGE  2
/* Voting.SingleObjectMaxType in SingleObjectMode (voting_mean_shift.cpp:80, 124-157; maxima_handler.h:45-46): "Default" / "None" (and
 * every type outside single-object mode) run the mean shift; the other three place ONE maximum per class at the cloud centroid */
#define ISMHIP_SOM_MEANSHIFT              0
#define ISMHIP_SOM_BANDWIDTH              1
#define ISMHIP_SOM_MODEL_RADIUS           2
#define ISMHIP_SOM_COMPLETE_VOTING_SPACE  3

#define ISMHIP_SHOT_DIM   352
#define ISMHIP_CSHOT_DIM 1344
#define ISMHIP_FPFH_DIM    33

typedef struct ismhip_ctx      ismhip_ctx;
typedef struct ismhip_cloud    ismhip_cloud;
typedef struct ismhip_codebook ismhip_codebook;

/* ---- context ----------------------------------------------------------------------------- */
int  ismhip_abi_version(void);
/* stream: a hipStream_t (as void*) the caller owns, or NULL to let the ctx create its own. */
int  ismhip_ctx_create(int device, void* stream, ismhip_ctx** out);
/* same, but every value of stream is taken literally: NULL is the device's default (null) stream. This is how a host
 * that already owns a stream (e.g. torch.cuda.current_stream()) orders its own work with the library's. */
int  ismhip_ctx_create_on_stream(int device, void* stream, ismhip_ctx** out);
int  ismhip_ctx_destroy(ismhip_ctx* ctx);
/* waits for the context's stream. Also the place where asynchronous caps surface: if find_maxima / hough3d_maxima had to drop
 * maxima since the last call (more than 128 per object and class, or 1024 per object), it returns ISMHIP_ERR_UNSUPPORTED once
 * (message in ismhip_last_error) and clears the condition -- the reference has no such caps, so this is never silent. */
int  ismhip_sync(ismhip_ctx* ctx);
const char* ismhip_last_error(const ismhip_ctx* ctx);
/* per-kernel device timers (hipEvent on the ctx stream). Enable, run, sync, then read.
 * name: "grid","lrf","shot352","cshot1344","fpfh33","knn","cast_votes","maxima". Returns accumulated
 * milliseconds and launch count since the last reset. */
int  ismhip_timers_enable(ismhip_ctx* ctx, int on);
int  ismhip_timers_reset(ismhip_ctx* ctx);
int  ismhip_timer_get(ismhip_ctx* ctx, const char* name, double* ms_out, int64_t* launches_out);

/* ---- search surface (replaces pcl::search::KdTree built at implicit_shape_model.cpp:823-831) ---
 * Takes the NaN-free surface cloud of n_obj objects as SoA and builds a per-object uniform grid
 * (y/z cell edge = cell_size, x cells three times finer; use 0.4 * min(Radius, ReferenceFrameRadius)) with the points counting-sorted
 * by cell. rgba may be NULL (needed only by cshot1344): packed as PCL does, 0x00RRGGBB. */
int  ismhip_cloud_create(ismhip_ctx* ctx, int n_obj, const uint32_t* pt_offsets_h,
                         const float* x, const float* y, const float* z,
                         const float* nx, const float* ny, const float* nz,
                         const uint32_t* rgba, float cell_size, ismhip_cloud** out);
int  ismhip_cloud_destroy(ismhip_ctx* ctx, ismhip_cloud* cloud);
/* Normals for clouds that come without them: ImplicitShapeModel::computeNormals, ConsistentNormalsMethod 2 (the default,
 * implicit_shape_model.cpp:1014-1018) -> NormalOrientation::processSHOTLRF (utils/normal_orientation.cpp:48-110): a SHOT
 * frame of radius NormalRadius at every point, normal = inverted z axis; NaN where the frame is invalid (< 5 neighbours).
 * The cloud may have been created with any normal arrays (their values are not read before this call); the outputs
 * (device, original point order, may alias the arrays given to ismhip_cloud_create) also replace the cloud's normals. */
int  ismhip_estimate_normals(ismhip_ctx* ctx, ismhip_cloud* cloud, float radius, float* nx_out, float* ny_out, float* nz_out);
/* ConsistentNormalsMethod 0 and 1 (implicit_shape_model.cpp:969-1011) -> pcl::NormalEstimationOMPWithEigVals
 * (third_party/pcl_normal_3d_omp_with_eigenvalues): PCA normal of the NormalRadius neighbourhood (>= 3 points, else NaN), flipped
 * towards the viewpoint. orientation 0: towards (0,0,0) (method 0); 1: away from the object's centroid (method 1). Outputs as
 * ismhip_estimate_normals. */
int  ismhip_estimate_normals_pca(ismhip_ctx* ctx, ismhip_cloud* cloud, float radius, int orientation, float* nx_out, float* ny_out, float* nz_out);
/* ImplicitShapeModel::filterNormals (implicit_shape_model.cpp:1034-1075): the points whose normal holds a NaN leave their cloud, order
 * kept, without the arrays leaving HBM. in / out: device SoA over all objects (out must not alias in; rgba may be NULL in both);
 * pt_offsets_h_out[n_obj+1] (host) receives the new ranges. The call synchronises. */
typedef struct ismhip_point_arrays { float *x, *y, *z, *nx, *ny, *nz; uint32_t* rgba; } ismhip_point_arrays;
int  ismhip_filter_normals(ismhip_ctx* ctx, int n_obj, const uint32_t* pt_offsets_h, const ismhip_point_arrays* in,
                           const ismhip_point_arrays* out, uint32_t* pt_offsets_h_out);
/* per-object centroid (features_shot.cpp:45-51) -> centroid_out[n_obj*3] */
int  ismhip_cloud_centroids(ismhip_ctx* ctx, const ismhip_cloud* cloud, float* centroid_out);
/* SingleObjectHelper::getModelRadius (voting/single_object_mode_helper.cpp:15-27): per object the largest distance of a (finite)
 * point from centroid[n_obj*3] (device, e.g. ismhip_cloud_centroids) -> radius_out[n_obj] (device) */
int  ismhip_cloud_radii(ismhip_ctx* ctx, const ismhip_cloud* cloud, const float* centroid, float* radius_out);

/* ---- local reference frames: Features::computeSHOTReferenceFrames (features/features.cpp:238-252)
 *      -> pcl::SHOTLocalReferenceFrameEstimationOMP (arithmetic as third_party/pcl_shot_na_lrf/shot_na_lrf.hpp:48-178
 *      with upstream's v.z sign rule). lrf9_out[nkp*9] row-major [x;y;z]; all-NaN when <5 neighbours. */
int  ismhip_shot_lrf(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                     const float* kpx, const float* kpy, const float* kpz,
                     float radius, float* lrf9_out);

/* ---- descriptors: FeaturesSHOT::iComputeDescriptors (features/features_shot.cpp:28-81) ------
 * desc_out[nkp*352]; NaN row when LRF non-finite or <5 neighbours. neighbour_count_out may be NULL
 * (else [nkp] number of radius neighbours, the M_k of the roofline model). */
int  ismhip_shot352(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                    const float* kpx, const float* kpy, const float* kpz,
                    const float* lrf9, float radius, float* desc_out, uint32_t* neighbour_count_out);
/* FeaturesCSHOT::iComputeDescriptors (features/features_cshot.cpp:28-103); kp_rgba = keypoint colours */
int  ismhip_cshot1344(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                      const float* kpx, const float* kpy, const float* kpz, const uint32_t* kp_rgba,
                      const float* lrf9, float radius, float* desc_out, uint32_t* neighbour_count_out);
/* FeaturesFPFH::iComputeDescriptors (features/features_fpfh.cpp:27-72) */
int  ismhip_fpfh33(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                   const float* kpx, const float* kpy, const float* kpz,
                   float radius, float* desc_out, uint32_t* neighbour_count_out);
/* ISMFeature::centerDist (features_shot.cpp:77): |keypoint - centroid(object)| -> out[nkp] */
int  ismhip_center_dist(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                        const float* kpx, const float* kpy, const float* kpz, float* out);

/* ---- keypoints (the step in front of the path): KeypointsVoxelGrid::iComputeKeypoints
 *      (keypoints/keypoints_voxel_grid.cpp:30-46) -> pcl::VoxelGrid<PointXYZRGB> with a cubic leaf. Per object the
 *      centroids (xyz, and rgb when rgba is given) of the occupied voxels in ascending voxel index, packed object
 *      after object into kx|ky|kz[|krgba] (device, `capacity` entries; the number of points always suffices);
 *      kp_offsets_h_out[n_obj+1] (host) receives the per-object ranges. Non-finite points are ignored. The call
 *      synchronises. */
int  ismhip_voxel_keypoints(ismhip_ctx* ctx, int n_obj, const uint32_t* pt_offsets_h,
                            const float* x, const float* y, const float* z, const uint32_t* rgba, float leaf,
                            uint32_t capacity, float* kx, float* ky, float* kz, uint32_t* krgba,
                            uint32_t* kp_offsets_h_out);

/* ---- feature filtering: Features::operator() drops non-finite LRFs (features.cpp:66-76),
 *      ImplicitShapeModel::removeNaNFeatures drops NaN descriptors (implicit_shape_model.cpp:1276-1308).
 *      Stable stream compaction of rows; keep_offsets_h_out[n_obj+1] (host) receives the new per-object
 *      ranges; src_index_out[nkp] (device) the source row of every kept row. The call synchronises. */
int  ismhip_compact_features(ismhip_ctx* ctx, int n_obj, const uint32_t* kp_offsets_h, int dim,
                             const float* desc, const float* lrf9,
                             const float* kpx, const float* kpy, const float* kpz,
                             float* desc_out, float* lrf9_out,
                             float* kpx_out, float* kpy_out, float* kpz_out,
                             uint32_t* src_index_out, uint32_t* keep_offsets_h_out);

/* The same filter for descriptor matrices written by ismhip_shot352 / ismhip_cshot1344 / ismhip_fpfh33, whose rows are NaN AS A WHOLE
 * (invalid frame, empty neighbourhood, zero norm): one element per row is tested instead of the matrix, and when nothing is dropped
 * *all_kept_out = 1, the *_out arrays are NOT written (the caller goes on with its input arrays; src_index_out, if given, is 0..nkp-1)
 * and keep_offsets_h_out = kp_offsets_h. Otherwise exactly as ismhip_compact_features. */
int  ismhip_compact_descriptor_rows(ismhip_ctx* ctx, int n_obj, const uint32_t* kp_offsets_h, int dim,
                                    const float* desc, const float* lrf9,
                                    const float* kpx, const float* kpy, const float* kpz,
                                    float* desc_out, float* lrf9_out,
                                    float* kpx_out, float* kpy_out, float* kpz_out,
                                    uint32_t* src_index_out, uint32_t* keep_offsets_h_out, int* all_kept_out);

/* ---- partial descriptors: Codebook::castVotes with UsePartialShot (codebook/codebook.cpp:416-475, mask :952-1036) keeps the
 *      histograms of some of the 32 SHOT signatures: dst[n_rows * n_cols] = src[:, cols_h] (cols_h host, ascending). */
int  ismhip_gather_columns(ismhip_ctx* ctx, int n_rows, int dim_in, const float* src, int n_cols, const int32_t* cols_h, float* dst);

/* ---- codebook: FlannHelper dataset (utils/flann_helper.cpp:21-70) + CodewordDistribution vote
 *      tables (codebook/codeword_distribution.cpp:73-144) + classSigmas (codebook.cpp:107,159-193).
 *      Rows of words_h are in getCodewords() order (ascending codeword id, codebook.cpp:856-859).
 *      Votes are CSR over words. All inputs are host arrays, copied once; the codebook stays resident. */
int  ismhip_codebook_create(ismhip_ctx* ctx, int n_words, int dim, const float* words_h,
                            const float* word_weight_h,          /* [n_words] Codeword::getWeight, may be NULL (=1) */
                            const uint32_t* vote_offsets_h,      /* [n_words+1] */
                            const float* vote_xyz_h,             /* [n_votes*3] vote in LRF coordinates */
                            const float* vote_weight_h,          /* [n_votes] learned centre weight, NULL = 1 */
                            const float* vote_class_weight_h,    /* [n_votes] statistical weight of the vote's class, NULL = 1 */
                            const uint32_t* vote_class_h,        /* [n_votes] */
                            const uint32_t* vote_instance_h,     /* [n_votes] */
                            const float* vote_bbox_quat_h,       /* [n_votes*4] (w,x,y,z), NULL = identity */
                            const float* vote_bbox_size_h,       /* [n_votes*3], NULL = 0 */
                            int n_classes, const float* class_sigma_h, /* [n_classes] variance per class id */
                            ismhip_codebook** out);
/* Codeword::getClassId per word (codebook/codeword.h:73-75; only meaningful for one-feature codewords). Default when not set:
 * the class of the word's first stored vote. Needed by ismhip_knn_rule only. */
int  ismhip_codebook_set_word_class(ismhip_ctx* ctx, ismhip_codebook* cb, const uint32_t* word_class_h);
int  ismhip_codebook_destroy(ismhip_ctx* ctx, ismhip_codebook* cb);
int  ismhip_codebook_max_votes_per_word(const ismhip_codebook* cb);
/* Diagnostic: leading rotated coordinates of the codebook's stage-1 search image (0 = none: the squared-L2 candidate stage runs on
 * all dimensions). Speed only -- every ismhip_knn answer is the exact functor minimum either way. energy_out (may be NULL): share
 * of the codebook's second moment those coordinates hold. */
int  ismhip_codebook_stage1_dims(const ismhip_codebook* cb, float* energy_out);
/* The same for the image the queries whose stage-1 proof failed are searched on again (0: all dimensions). */
int  ismhip_codebook_stage2_dims(const ismhip_codebook* cb, float* energy_out);

/* ---- activation: ActivationStrategyKNN::activateKNN (activation_strategy/activation_strategy_knn.h:41-126)
 *      with FLANNExactMatch semantics (SearchParams(-1)): exact k nearest codewords, ascending distance,
 *      ties -> lowest row. idx_out[nq*k] (row in words, -1 when n_words < k), dist_out[nq*k] = the FLANN
 *      functor value (utils/distance.cpp:33-52), recomputed by direct summation for the winners. */
int  ismhip_knn(ismhip_ctx* ctx, const ismhip_codebook* cb, int metric, int nq, const float* q,
                int k, int32_t* idx_out, float* dist_out);
/* distance-ratio test of activateKNN (:74-85), k must be 1: needs the 2-NN; idx -> -1 when d1/d2 > threshold */
int  ismhip_knn_ratio(ismhip_ctx* ctx, const ismhip_codebook* cb, int metric, int nq, const float* q,
                      float ratio_threshold, int32_t* idx_out, float* dist_out);

/* ActivationStrategyKnnRule::activateKNN at detection time (activation_strategy/activation_strategy_knn_rule.h:41-152): exact
 * 3-NN, then the class-consistency rules over (c1,c2,c3) with the ratio tests d1/d3 and d1/d2; idx_out[nq] = accepted codeword
 * row (k1 or k2) or -1, dist_out[nq] = its functor distance. (At training time the rule is plain 1-NN: use ismhip_knn.) */
int  ismhip_knn_rule(ismhip_ctx* ctx, const ismhip_codebook* cb, int metric, int nq, const float* q,
                     float ratio_threshold, int32_t* idx_out, float* dist_out);

/* ---- vote casting: Codebook::castVotes second loop + CodewordDistribution::castVotes/castVote
 *      (codebook.cpp:541-554, codeword_distribution.cpp:73-167), sink = Voting::vote (voting/voting.cpp:58-77).
 *      Vote slot of (feature f, activation j, stored vote v) = (f*k + j)*maxv + v with
 *      maxv = ismhip_codebook_max_votes_per_word; a slot that casts no vote has class = -1.
 *      All outputs are SoA of n_slots = nq*k*maxv entries. */
int  ismhip_cast_votes(ismhip_ctx* ctx, const ismhip_codebook* cb, uint32_t weight_flags,
                       int nq, const float* lrf9, const float* kpx, const float* kpy, const float* kpz,
                       int k, const int32_t* idx, const float* dist,
                       float* vote_pos_out,      /* [n_slots*3] */
                       float* vote_weight_out,   /* [n_slots] */
                       int32_t* vote_class_out,  /* [n_slots], -1 = no vote */
                       int32_t* vote_instance_out,
                       int32_t* vote_codeword_out,
                       float* vote_bbox_quat_out,/* [n_slots*4], may be NULL */
                       float* vote_bbox_size_out /* [n_slots*3], may be NULL */);

/* ---- maxima: Voting::findMaxima + VotingMeanShift::iFindMaxima + MaximaHandler
 *      (voting/voting.cpp:79-328,436-462; voting_mean_shift.cpp:39-177,201-481; maxima_handler.cpp:51-157) */
typedef struct ismhip_maxima_params {
    int   n_classes;
    const float* class_bandwidth_h; /* [n_classes] MaximaHandler::getSearchDistForClass; NULL -> bandwidth for all */
    float bandwidth;                /* Voting.Bandwidth */
    float threshold;                /* Voting.Threshold */
    int   max_iter;                 /* Voting.MaxIter */
    int   kernel;                   /* ISMHIP_KERNEL_* */
    int   suppression;              /* ISMHIP_SUPPRESS_* */
    int   min_votes_threshold;      /* Voting.MinVotesThreshold */
    float min_threshold;            /* Voting.MinThreshold (negative = relative to best) */
    int   best_k;                   /* Voting.BestK (<=0: all) */
    int   max_maxima;               /* capacity of the output per object */
    int   max_filter;               /* Voting.MaxFilterType: ISMHIP_MAXFILTER_NONE | _SIMPLE | _MERGE (MaximaHandler::filterMaxima,
                                       maxima_handler.cpp:272-440); not applied in single-object mode: pass NONE there (voting.cpp:262-268) */
    /* ---- ABI 4 (zero-initialise the struct: all of these are optional) */
    const float* vote_bbox_quat;    /* device [n_slots*4] (w,x,y,z), the vote_bbox_quat_out of ismhip_cast_votes: Voting.AverageRotation
                                       (voting.cpp:210-215, Utils::quatWeightedAverage utils.cpp:617-665; see DESIGN.md §7 for the
                                       eigenvector choice); NULL = off */
    float* max_bbox_quat_out;       /* device [n_obj*max_maxima*4]; required iff vote_bbox_quat is given */
    int   single_object_max_type;   /* ISMHIP_SOM_* */
    const float* object_centroid;   /* device [n_obj*3]: ismhip_cloud_centroids of the objects' clouds (ISMHIP_SOM_BANDWIDTH and up) */
    const float* object_radius;     /* device [n_obj]: ismhip_cloud_radii (ISMHIP_SOM_MODEL_RADIUS) */
} ismhip_maxima_params;

/* slot_offsets_h[n_obj+1]: vote-slot range of each object. Outputs per object o, maximum m (sorted by
 * weight, descending): index o*max_maxima + m. n_maxima_out[n_obj]. class_score_out[n_obj*n_classes] =
 * best normalised weight per class (0 when the class has no maximum) — the record the multi-GPU
 * all-gather exchanges. */
int  ismhip_find_maxima(ismhip_ctx* ctx, int n_obj, const uint32_t* slot_offsets_h,
                        const float* vote_pos, const float* vote_weight, const int32_t* vote_class,
                        const int32_t* vote_instance, const float* vote_bbox_size /* may be NULL */,
                        const ismhip_maxima_params* params,
                        int32_t* n_maxima_out, float* max_pos_out, float* max_weight_out,
                        int32_t* max_class_out, int32_t* max_instance_out, float* max_instance_weight_out,
                        float* max_bbox_size_out /* may be NULL */, int32_t* max_n_votes_out,
                        float* class_score_out);

/* ---- training: Codebook::activate (codebook/codebook.cpp:64-368): exact kNN activation of every training feature in the
 *      codebook, class sigma^2 (:94-193), K = 1 clean-up (:201-224), vote = rotateInto(centre - keypoint, LRF)
 *      (codeword_distribution.cpp:37-71), CodewordDistribution::computeWeights (:169-243) and the statistical class weights
 *      term1 * term2 * term3 (:226-368, including m_term3 being keyed by class only).
 *      The codewords are the rows of `codewords` (device [n_codewords * dim]: the cluster centres of ismhip_kmeans,
 *      implicit_shape_model.cpp:445-475), or, with codewords == NULL, the training features themselves (clustering_none.cpp:25-35).
 *      desc / lrf9 / kp* are device arrays of the n training features in CLASS-MAJOR order (classes ascending, models and
 *      features in the order the reference iterates them); feat_*_h are host arrays ([n], centre [n*3] = the model's bounding-box
 *      centre). Outputs are HOST arrays with m = number of codewords: word_src_out[m] (row of `codewords` behind every kept
 *      codeword, ascending = codeword order), vote_offsets_out[m+1] (CSR), vote_feature_out / vote_weight_out /
 *      vote_class_weight_out [n*k], vote_xyz_out[n*k*3], class_sigma_out[n_classes]. The call synchronises. k <= 16;
 *      a codeword with more than 32768 votes is refused (ISMHIP_ERR_UNSUPPORTED). */
int  ismhip_train_activate(ismhip_ctx* ctx, int metric, int n, int dim, const float* desc, const float* lrf9,
                           const float* kpx, const float* kpy, const float* kpz,
                           const uint32_t* feat_class_h, const uint32_t* feat_model_h, const float* feat_center_h,
                           int n_codewords, const float* codewords /* device, may be NULL */,
                           int k, int clean_up_single_vote, int n_classes,
                           int32_t* n_words_out, uint32_t* word_src_out, uint32_t* vote_offsets_out, uint32_t* vote_feature_out,
                           float* vote_xyz_out, float* vote_weight_out, float* vote_class_weight_out, float* class_sigma_out);

/* ---- k-means codebook clustering: ClusteringKMeans::cluster (clustering/clustering_kmeans.h:53-131) =
 *      flann::hierarchicalClustering with branching == the cluster count (one level of Lloyd k-means: centre chooser, then
 *      [means -> reassign, ties to the lowest centre -> refill empty clusters] until nothing moves or max_iterations), followed by
 *      the nearest centre of every feature. FLANN is EXTERNAL and draws from rand(): the random draws, the integer form of the
 *      k-means++ sampling and of the means, and the exact final search are this library's own definitions (csrc/kmeans.hip,
 *      restated in oracle/; parity with the reference unpinned). desc: device [n * dim]. n_clusters is clipped to n.
 *      centers_out: device [n_clusters * dim]; assign_out: device [n] (row of centers_out); dist_out: device [n] or NULL (functor
 *      distance to that centre); *n_clusters_out <= n_clusters (fewer when the distinct points run out). Synchronises. */
#define ISMHIP_CENTERS_RANDOM   0
#define ISMHIP_CENTERS_GONZALES 1
#define ISMHIP_CENTERS_KMEANSPP 2
int  ismhip_kmeans(ismhip_ctx* ctx, int metric, int n, int dim, const float* desc, int n_clusters, int max_iterations,
                   int centers_init, unsigned long long seed, float* centers_out, int32_t* assign_out, float* dist_out,
                   int32_t* n_clusters_out, int32_t* iterations_out /* may be NULL */);

/* ---- discrete Hough space: VotingHough3D::iFindMaxima (voting/voting_hough_3d.cpp:33-95) over pcl::recognition::HoughSpace3D
 *      (bins ceil((max-min)/bin) per axis; trilinear voteInt; findMaxima(-RelThreshold): bins >= rel * max with no strictly
 *      greater 26-neighbour, in ascending bin index) followed by the same Voting::findMaxima post-processing as
 *      ismhip_find_maxima (same outputs). The reference makes the bins cubic: edge = 2 * MaximaHandler::getSearchDistForClass,
 *      i.e. BinSize[0] for BinOrBandwidthType "Config" (voting_hough_3d.cpp:46-48). */
typedef struct ismhip_hough_params {
    int   n_classes;
    float min_coord[3];             /* Voting.MinCoord */
    float max_coord[3];             /* Voting.MaxCoord */
    float bin_size;                 /* Voting.BinSize[0] */
    const float* class_bin_h;       /* [n_classes] per-class bin edge; NULL -> bin_size for all */
    int   use_interpolation;        /* Voting.UseInterpolation */
    float rel_threshold;            /* Voting.RelThreshold */
    int   min_votes_threshold;      /* Voting.MinVotesThreshold */
    float min_threshold;            /* Voting.MinThreshold (negative = relative to best) */
    int   best_k;                   /* Voting.BestK (<=0: all) */
    int   max_maxima;               /* capacity of the output per object */
    int   max_filter;               /* as ismhip_maxima_params.max_filter; the radius is bin_size / 2 (voting_hough_3d.cpp:45) */
    /* ---- ABI 4 (zero-initialise the struct) */
    const float* vote_bbox_quat;    /* Voting.AverageRotation, as in ismhip_maxima_params */
    float* max_bbox_quat_out;
} ismhip_hough_params;
int  ismhip_hough3d_maxima(ismhip_ctx* ctx, int n_obj, const uint32_t* slot_offsets_h,
                           const float* vote_pos, const float* vote_weight, const int32_t* vote_class,
                           const int32_t* vote_instance, const float* vote_bbox_size /* may be NULL */,
                           const ismhip_hough_params* params,
                           int32_t* n_maxima_out, float* max_pos_out, float* max_weight_out,
                           int32_t* max_class_out, int32_t* max_instance_out, float* max_instance_weight_out,
                           float* max_bbox_size_out /* may be NULL */, int32_t* max_n_votes_out,
                           float* class_score_out);

#ifdef __cplusplus
}
#endif
#endif /* ISMHIP_H_ */
