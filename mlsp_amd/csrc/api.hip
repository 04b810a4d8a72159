// extern "C" entry points of libmlsp_hip.so (declared in include/mlsp_hip.h).  Host-side sequencing
// only: argument checks, workspace carving, kernel launches on the caller's stream.  No allocation,
// no synchronisation, no global state (hipGraph-capturable).
#include "common.h"
#include "../../include/mlsp_hip.h"

// ---- launchers implemented in the other translation units -----------------------------------
int launch_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                float* C, int ldc, const float* bias, const float* gbias, int rows_per_group, float* slab, size_t slab_floats,
                double* stat_part = nullptr, const float* sel_gamma = nullptr, float* sel_val = nullptr, int* sel_row = nullptr,
                bool accumulate = false, const GemmXf* xf = nullptr, int stat_ld = 0, const GemmGroups* grp = nullptr, const GemmBs* bs = nullptr,
                const GemmDy* dy = nullptr);
bool gemm_dy_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb);
int launch_bn_bwd_finalize_coef(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                float* dbeta, float* coef);
int launch_bn_dy_gbias(hipStream_t st, const float* Y, int G, int rows_per_group, int C, const double* stats, int panel_rows, const float* coef,
                       float* scratch, float* out, const float* ysum);
int launch_bn_bwd_finalize_coef_groups(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                       float* dbeta, float* coef, const float* gys, int ppg, int rows, float* gout);
int launch_bn_finalize_groups(hipStream_t st, const double* part, int nparts, double count, int C, const float* gamma,
                              const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* scale,
                              float* shift, float* save_mean, float* save_invstd, float* gsum, int ppg);
bool gemm_xf_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int which);
int launch_gemm_mx(hipStream_t st, bool ta, bool tb, int M, int N, int K, const void* A, int a_bf16, int lda, const void* B, int b_bf16,
                   int ldb, void* C, int c_bf16, int ldc, const float* bias, const float* gbias, int rows_per_group, float* slab,
                   size_t slab_floats, double* stat_part, bool accumulate);
int launch_bn_act_fwd_b16(hipStream_t st, const void* Y, void* Z, int rows, int C, const float* scale, const float* shift, int act,
                          float slope, float p_drop, uint64_t seed);
int launch_bn_act_bwd_b16(hipStream_t st, const void* dZ, const void* Y, void* dY, int M, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, int training, int act, float slope, float p_drop, uint64_t seed,
                          double* part, float* dgamma, float* dbeta, float* mean_dz, float* mean_dzy);
int launch_colsum_groups_b16(hipStream_t st, const void* X, int G, int rows_per_group, int C, float* out, float* scratch);
int launch_colsel_panels(hipStream_t st, const float* pv, const int* pr, const float* gamma, int B, int N, int C, int panel_rows,
                         float* ysel, int* arg, const float* bn, int act, float slope, float* out);
int gemm_panel_rows(int M, int N, int K);
int launch_xf_materialize(hipStream_t st, const float* X, int ldx, int M, int C, const GemmXf& xf, float* out);   // multi.hip
int gemm_stat_parts(int M, int N, int K);
size_t gemm_slab_floats(int M, int N, int K);
int launch_knn(hipStream_t st, const float* x, int ld, int B, int N, int C, int k, int* idx, float* xx_ws, void* planes, size_t plane_bytes);
int launch_compose_fwd(hipStream_t st, const float* Wa, const float* ba, const float* Wb, const float* bb, int Cm, int Ci, int Co, float* W, float* b);
int launch_compose_bwd(hipStream_t st, const float* dW, const float* db, const float* Wa, const float* ba, const float* Wb, int Cm, int Ci, int Co,
                       float* dWa, float* dba, float* dWb);
int launch_skinny_bwd_pair(hipStream_t st, const float* G, int ldg, const float* W, int ldw, const float* X, int ldx, float* dX, int lddx,
                           float* dW, int M, int Cin, int Cout, int dx_accumulate = 0);
int launch_wt_vec_neg_scale_rows(hipStream_t st, const float* W, int ldw, const float* v, const float* rowscale, int Cout, int Cin, float* negr,
                                 float* Wb);
int launch_bn_bwd_finalize_coef_z(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                  float* dbeta, float* coef, float* zero_vec);
int launch_skinny_bn_bwd_z(hipStream_t st, const float* dZ, const float* Y, float* dY, int M, int C, const float* bn_save, int training,
                           int act, float slope, float p_drop, uint64_t seed, float* dgamma, float* dbeta, float* zero_vec);
void gemm_unfold_request(float* dW);
void bn_bound_request(float* out);
float* amax_offered_output(const float* out, long rows, int cols, int ld, int need);
float* amax_reserve(const float* X, long rows, int cols, int ld);
bool gemm_unfold_take();
void bn_zero_vec_request(float* v);
bool bn_zero_vec_take();
int launch_build_wd_eval(hipStream_t st, const float* W, int Cout, int C, float* Wd, int Ca, const float* ga, const float* ba, const float* rma,
                         const float* rva, float* sva, int Cb, const float* gb, const float* bb, const float* rmb, const float* rvb, float* svb,
                         float eps);
bool knn6_supported(int B, int N, int C, int k);
bool knn6w_supported(int B, int N, int C, int k);
size_t knn6_plane_bytes(int P, int C);
int launch_knn_reverse(hipStream_t st, const int* idx, int B, int N, int k, int* rev_off, int* rev_ent);
int bn_stat_parts(int M);
int bn_parts_max(int M);
int launch_skinny_linear_bn_act(hipStream_t st, const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout,
                                const float* bias, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                float momentum, float eps, int training, int act, float slope, float p_drop, uint64_t seed, float* Y,
                                float* Z, float* bn_save);
int launch_skinny_bn_bwd(hipStream_t st, const float* dZ, const float* Y, float* dY, int M, int C, const float* bn_save, int training,
                         int act, float slope, float p_drop, uint64_t seed, float* dgamma, float* dbeta);
int launch_group_reverse(hipStream_t st, const int* idx, int B, int S, int N, int k, int* rev_off, int* rev_ent);
int launch_fps(hipStream_t st, const float* xyz, int ldx, int B, int N, int S, const int* start, int* out);
int launch_ball_query(hipStream_t st, const float* xyz, int ldx, const float* q, int ldq, int B, int N, int S, float r2, int nsample,
                      int* idx);
int launch_sa_group_fwd(hipStream_t st, const float* xyz, int ldx, const float* feat, int D, const float* q, int ldq, const int* idx,
                        int B, int N, int S, int ns, float* G);
int launch_sa_group_bwd(hipStream_t st, const float* dG, int ldg, int col, int D, const int* rev_off, const int* rev_ent, int B, int N, int S, int ns,
                        float* dfeat);

int launch_collapse_to_point(hipStream_t st, float* X, int B, int N, const int* choice, const float* u, const float* noise, float r2,
                             int min_pts, float* mask, int* chosen);
int launch_knn_query(hipStream_t st, const float* ref, int ldr, int Nr, const float* qry, int ldq, int Nq, int B, int C, int k, int* idx,
                     float* dist);
int launch_interp3_fwd(hipStream_t st, const float* feat, const int* idx, const float* dist, int B, int N, int S, int D, float* out);
int launch_interp3_bwd(hipStream_t st, const float* dout, const float* dist, const int* rev_off, const int* rev_ent, int B, int N, int S,
                       int D, float* dfeat);
int launch_transform3_fwd(hipStream_t st, const float* x, const float* T, int B, int N, float* out);
int launch_transform3_bwd(hipStream_t st, const float* x, const float* T, const float* dout, int B, int N, float* dx, float* dT);
int launch_region_assign(hipStream_t st, const float* X, int B, int C, int N, const float* thr, int n, float clip, int* Y);
int launch_scan_select(hipStream_t st, const float* X, int B, int N, int C, const double* R, int pixel, float* Xs, float* mask);
int launch_deform_regions(hipStream_t st, float* X, int B, int C, int N, const int* regions, const int* order, int nreg, const float* lookup,
                          const float* noise, int min_pts, int groups, float* mask);
int launch_bn_act_bwd_partials_vec(hipStream_t st, const float* dZ, const float* Y, int M, int C, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, int act, float slope, double* part);
int bn_vec_parts(int M);
int launch_colstats(hipStream_t st, const float* Y, int M, int C, int ld, double* part);
int launch_bn_finalize(hipStream_t st, const double* part, int nparts, double count, int C, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* scale,
                       float* shift, float* save_mean, float* save_invstd);
int launch_bn_eval_prepare(hipStream_t st, int C, const float* gamma, const float* beta, const float* run_mean,
                           const float* run_var, float eps, float* scale, float* shift, float* save_mean, float* save_invstd);
int launch_bn_act_fwd(hipStream_t st, const float* Y, float* Z, size_t rows, int C, const float* scale, const float* shift,
                      int act, float slope, float p_drop, uint64_t seed);
int launch_bn_act_bwd(hipStream_t st, const float* dZ, const float* Y, float* dY, int M, int C, const float* scale,
                      const float* shift, const float* mean, const float* invstd, int training, int act, float slope,
                      float p_drop, uint64_t seed, double* part, float* dgamma, float* dbeta, float* mean_dz, float* mean_dzy,
                      float* gpart = nullptr, int rows_per_group = 0, int* gpart_slabs = nullptr, const double* pre_stats = nullptr, int pre_parts = 0);
int gemm_bs_parts(int M, int N, int K, int lda, int ldb, int ldc);
int thin_bs_parts(int M, int N, int K);
int launch_colsum_groups_fin(hipStream_t st, const float* scratch, int G, int C, int slabs, float* out);
int launch_colsum_groups(hipStream_t st, const float* X, int G, int rows_per_group, int C, float* out, float* scratch = nullptr);
int launch_colsum(hipStream_t st, const float* X, int M, int C, double* part, float* out);
int launch_colmax_fwd(hipStream_t st, const float* Z, int B, int N, int C, float* out, int* arg);
int launch_colmax_bwd(hipStream_t st, const float* dOut, const int* arg, int B, int N, int C, float* dZ);
int launch_segmax_fwd(hipStream_t st, const float* Z, int P, int k, int C, float* out, uint8_t* argk);
int launch_segmax_bwd(hipStream_t st, const float* dOut, const uint8_t* argk, int P, int k, int C, float* dZ);
int edge_reduce_parts(int P);
int edge_bwd_reduce_parts(int P, int Cout, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f,
                          int lddo, int ldo);
int launch_sa_fold_fwd(hipStream_t st, const float* u, const float* w, const int* idx, int B, int N, int S, int ns, int C, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, float momentum, float eps, int training, float* Z, float* bn_save,
                       double* part);
int launch_sa_fold_bwd(hipStream_t st, const float* dZ, const float* u, const float* w, const int* idx, const int* rev_off, const int* rev_ent,
                       int B, int N, int S, int ns, int C, const float* bn_save, int training, float* du, float* dw, float* dgamma,
                       float* dbeta, double* part, float* mean_dz, float* mean_dzy, const int* rev_cnt, const int* pad_cnt);
int launch_group_reverse_compact(hipStream_t st, const int* idx, int B, int S, int N, int k, int* rev_off, int* rev_cnt, int* rev_ent,
                                 int* pad_cnt);
int sa_fold_parts(long E);
int launch_segsel_act_fwd(hipStream_t st, const float* Y, int G, int k, int C, const float* scale, const float* shift, int act, float slope,
                          float* out, float* ysel, uint8_t* argk);
int launch_segsel_bwd_apply(hipStream_t st, const float* dOut, const float* Y, const float* ysel, const uint8_t* argk, size_t M, int k, int C,
                            const float* bn, const float* m1, const float* m2, int act, float slope, float* dY);
int launch_build_wd(hipStream_t st, const float* W, int Cout, int C, float* Wd, float* amax = nullptr);
bool build_wd_leaves_bound(int Cout, int C);
int launch_unbuild_wd(hipStream_t st, const float* dWd, int Cout, int C, float* dW);
int launch_edge_reduce(hipStream_t st, const float* uv, const int* idx, const float* gamma, int P, int N, int Cout, int k,
                       float* msel, uint8_t* argsel, float* s1, double* part, int* nparts_used);
int launch_edge_select_act(hipStream_t st, const float* msel, const float* uv, int P, int Cout, const float* scale,
                           const float* shift, int act, float slope, float* out, int ldo);
int launch_edge_bwd_reduce(hipStream_t st, const float* dOut, const float* out, const float* msel, const float* uv, int P,
                           int Cout, const float* mean, const float* invstd, int act, float slope, double* part, int lddo, int ldo,
                           float* duv_amax);
bool edge_bwd_leaves_duv_bound(const float* dOut, const float* out, const float* msel, const float* uv, const float* s1, const uint8_t* argsel,
                               int Cout, const float* scale, const float* mean, const float* invstd, const float* gz, const float* duv,
                               int lddo, int ldo);
int launch_edge_bwd_point(hipStream_t st, const float* dOut, const float* out, const float* uv, const float* s1, int P,
                          int Cout, int k, const float* scale, const float* mean, const float* invstd, const float* mean_dz,
                          const float* mean_dzy, int act, float slope, float* gz, float* duv, int lddo, int ldo, float* duv_amax);
int launch_edge_bwd_gather(hipStream_t st, const float* gz, const uint8_t* argsel, const float* uv, const int* rev_off,
                           const int* rev_ent, int P, int N, int Cout, const float* scale, const float* mean,
                           const float* invstd, const float* mean_dz, const float* mean_dzy, float* duv, float* duv_amax);
int launch_graph_feature_fwd(hipStream_t st, const float* x, const int* idx, int P, int N, int C, int k, float* F);
int launch_graph_feature_bwd(hipStream_t st, const float* dF, const int* rev_off, const int* rev_ent, int P, int N, int C,
                             int k, float* dx);
int launch_chamfer_fwd(hipStream_t st, const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                       float* per_cloud, int* argA, int* argB, float* loss);
int launch_chamfer_bwd(hipStream_t st, const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                       const float* per_cloud, const int* argA, const int* argB, const float* gout, float* dpred);
int launch_chamfer_dir_fwd(hipStream_t st, const float* p1, const float* p2, const float* mc, int B, int N, float* per_cloud, int* arg,
                           float* loss);
int launch_chamfer_dir_bwd(hipStream_t st, const float* p1, const float* p2, const float* mc, int B, int N, const float* per_cloud,
                           const int* arg, const float* gout, float* dp1, float* dp2);
int launch_normal_loss_fwd(hipStream_t st, const float* pred, const float* gt, const float* w, int P, float weight,
                           double* part, float* out);
int launch_normal_loss_bwd(hipStream_t st, const float* pred, const float* gt, const float* w, int P, float weight,
                           const float* fwd_out, const float* gout, float* dpred);
int launch_density_tail_fwd(hipStream_t st, const float* logits, const float* w, int P, int nc, float* pvec, float* dens);
int launch_density_tail_bwd(hipStream_t st, const float* pvec, const float* w, const float* dp, const float* dd, int P, int nc,
                            float* dlogits);
int launch_density_loss_fwd(hipStream_t st, const float* pvec, const float* dens, const float* tvec, const float* target,
                            const float* m, int P, int nc, float dweight, double* part, float* out);
int launch_density_loss_bwd(hipStream_t st, const float* pvec, const float* dens, const float* tvec, const float* target,
                            const float* m, int P, int nc, float dweight, const float* fwd_out, const float* gkl,
                            const float* gmae, float* dp, float* dd);
int launch_bn_bwd_finalize(hipStream_t st, const double* part, int nparts, double count, int C, float* dgamma, float* dbeta,
                           float* mean_dz, float* mean_dzy);

int launch_slab_reduce(hipStream_t st, const float* slab, float* C, int M, int N, int ldc, int nsplit);
int tnet_grid(int ntiles);
int tnet_points_per_tile(int k);
int tnet_fwd_parts(int B, int N, int k);
int launch_tnet_edge_fwd(hipStream_t st, const float* uv, const int* idx, const float* bn1, const float* W2, const float* gamma2,
                         int P, int N, int k, float slope, float* zsel, uint8_t* argsel, double* part);
int launch_tnet_out(hipStream_t st, const float* zsel, const float* bn2, int P, float slope, float* out);
int launch_tnet_bwd_reduce(hipStream_t st, const float* dT, const float* T, const float* zsel, const float* bn2, int P,
                           float slope, double* part);
int launch_tnet_bwd_g(hipStream_t st, const float* dT, const float* T, const float* bn2, const float* mean_dz,
                      const float* mean_dzy, int P, float slope, float* g, float* coef);
size_t tnet_bwd_scratch_floats(int ntiles);
int launch_tnet_edge_bwd(hipStream_t st, const float* uv, const int* idx, const float* bn1, const float* W2, const float* bn2,
                         const float* g, const uint8_t* argsel, const float* coef, int P, int N, int k, float slope, float* dhp,
                         float* scratch, double* part1, float* dW2, int* nparts);
size_t tnet_w1_moment_doubles(int P, int k);
int launch_tnet_bwd_w1_moments(hipStream_t st, const float* dhp, const int* idx, const int* rev_off, const float* x, int ldx, const float* W1,
                               const float* bn1, const float* m1, const float* m2, int P, int N, int k, int C, double* scratch, float* dW1);
int launch_tnet_edge_bwd2(hipStream_t st, const float* dhp, const float* uv, const float* s1, const float* bn1, const float* m1,
                          const float* m2, const int* rev_off, const int* rev_ent, int P, int N, int k, float* duv);

int launch_colsel(hipStream_t st, const float* Y, const float* gamma, int B, int N, int C, float* ysel, int* arg);
int launch_colsel_out(hipStream_t st, const float* ysel, const float* bn, int B, int C, int act, float slope, float* out);
int launch_colmax_bwd_coef(hipStream_t st, const float* dOut, const float* out, const float* ysel, const float* bn, int B, int C,
                           double count, int act, float slope, int training, float* g, float* coef, float* dgamma, float* dbeta);
int launch_scale_rows(hipStream_t st, const float* W, int ldw, const float* rowscale, int Cout, int Cin, float* Wb);
int launch_wt_vec_neg(hipStream_t st, const float* W, int ldw, const float* v, int Cout, int Cin, float* negr);
int launch_colmax_gather_rows(hipStream_t st, const float* g, const int* arg, const float* X, int ldx, int B, int N, int Cout,
                              int Cin, float* S);
int launch_colmax_dw(hipStream_t st, const float* S, const float* WG, const float* sx, const float* coef, const float* bn, int Cout,
                     int Cin, float* dW);
int launch_colmax_scatter_rows(hipStream_t st, const float* g, const int* arg, const float* W, int ldw, int B, int N, int Cout,
                               int Cin, float* dX, int lddx);

int launch_radius_count(hipStream_t st, const float* x, int ld, int B, int N, float radius, int max_nn, int* count);
int launch_knn_normals(hipStream_t st, const float* x, int ld, const int* idx, int B, int N, int k, float* normals);

#define CHECK(x) do { int _r = (x); if (_r != MLSP_OK) return _r; } while (0)
#define SLAB_BOUND_FLOATS ((size_t)16 << 20)   /* 64 MiB of fp32: bound on any split-K slab (gemm_pick_split) */

#include <mutex>
#include <unordered_map>
hipError_t mlsp_lds_limit(const void* fn, size_t lds) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, size_t> raised;          // (device, kernel) -> limit already set
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)(uintptr_t)fn * 64u + (uint64_t)(dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    auto it = raised.find(key);
    if (it != raised.end() && it->second >= lds) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) raised[key] = lds;
    return e;
}

extern "C" {

int mlsp_abi_version(void) { return MLSP_ABI_VERSION; }

const char* mlsp_strerror(int code) {
    switch (code) {
        case MLSP_OK: return "ok";
        case MLSP_ERR_ARG: return "mlsp: bad argument (null pointer, non-positive or unsupported shape)";
        case MLSP_ERR_WORKSPACE: return "mlsp: workspace too small (see mlsp_workspace_bytes)";
        case MLSP_ERR_UNSUPPORTED: return "mlsp: unsupported size for this kernel";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "mlsp: unknown error";
    }
}

size_t mlsp_workspace_bytes(int rows, int cin, int cout) {
    size_t r = rows > 0 ? rows : 1, ci = cin > 0 ? cin : 1, co = cout > 0 ? cout : 1;
    size_t act = 3 * r * co * sizeof(float) + r * sizeof(float)             // gz + duv (edgeconv bwd) / dY (mlp bwd) / xx
               + r * ci * sizeof(float);                                    // a chained layer's activated input, shapes outside the fused path
    size_t parts = (r / 64 + 2) * 2 * co * sizeof(double);                  // stat partials
    size_t wts = 4 * co * ci * sizeof(float) * 2 + 8 * co * sizeof(float);  // Wd, dWd, coefficient vectors
    return act + parts + wts + SLAB_BOUND_FLOATS * sizeof(float) + (32 << 20) + MLSP_AMAX_TAIL_BYTES;       // + fixed-size per-workgroup partial slabs + the tail PREC_SCOPE reserves
}

int mlsp_gemm_f32(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    Workspace w(ws, ws_bytes);
    size_t sf = gemm_slab_floats(M, N, K);
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (sf && !slab) sf = 0;   // launcher falls back to a single pass
    return launch_gemm(st, ta != 0, tb != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, 0, slab, sf);
}

int mlsp_knn_f32(const float* x, int ldx, int B, int N, int C, int k, int32_t* idx, int32_t* rev_off, int32_t* rev_ent,
                 void* ws, size_t ws_bytes, mlsp_stream_t st) {
    if (B <= 0 || N <= 0) return MLSP_ERR_ARG;
    Workspace w(ws, ws_bytes);
    float* xx = w.take<float>((size_t)B * N);
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    const size_t pbytes = (knn6_supported(B, N, C, k) || knn6w_supported(B, N, C, k)) ? knn6_plane_bytes(B * N, C) : 0;      // bf16 hi / lo images of the points (knn6.hip)
    char* planes = pbytes ? w.take<char>(pbytes) : nullptr;          // (a workspace too small for them keeps the call on the v5 kernels)
    {   // bench.py roofline_kernels: C <= 4 priced against HBM (compulsory bytes), C = 64 / 128 against the fp32 matrix peak
        const int cls = C <= 4 ? MLSP_PROF_KNN_C3 : C == 64 ? MLSP_PROF_KNN_C64 : C == 128 ? MLSP_PROF_KNN_C128 : 0;
        const int tok = prof_cls_begin(st, cls);
        const int rc = launch_knn(st, x, ldx, B, N, C, k, idx, xx, planes, planes ? pbytes : 0);
        prof_cls_end(st, tok, C <= 4 ? (double)B * N * (C + k) * 4.0 : 2.0 * B * N * (double)N * C);
        if (rc != MLSP_OK) return rc;
    }
    if (rev_off) CHECK(launch_knn_reverse(st, idx, B, N, k, rev_off, rev_ent));
    return MLSP_OK;
}

int mlsp_knn_reverse(const int32_t* idx, int B, int N, int k, int32_t* rev_off, int32_t* rev_ent, mlsp_stream_t st) {
    return launch_knn_reverse(st, idx, B, N, k, rev_off, rev_ent);
}

int mlsp_region_assign_f32(const float* X, int B, int C, int N, const float* thr, int n, float clip, int32_t* regions, mlsp_stream_t st) {
    return launch_region_assign(st, X, B, C, N, thr, n, clip, regions);
}
int mlsp_deform_regions_f32(float* X, int B, int C, int N, const int32_t* regions, const int32_t* order, int nreg, const float* lookup,
                            const float* noise, int min_pts, int groups, float* mask, mlsp_stream_t st) {
    return launch_deform_regions(st, X, B, C, N, regions, order, nreg, lookup, noise, min_pts, groups, mask);
}
int mlsp_knn_query_f32(const float* ref, int ldr, int Nr, const float* qry, int ldq, int Nq, int B, int C, int k, int32_t* idx, float* dist,
                       mlsp_stream_t st) {
    return launch_knn_query(st, ref, ldr, Nr, qry, ldq, Nq, B, C, k, idx, dist);
}
int mlsp_interp3_fwd_f32(const float* feat, const int32_t* idx, const float* dist, int B, int N, int S, int D, float* out, mlsp_stream_t st) {
    return launch_interp3_fwd(st, feat, idx, dist, B, N, S, D, out);
}
int mlsp_interp3_bwd_f32(const float* dout, const float* dist, const int32_t* rev_off, const int32_t* rev_ent, int B, int N, int S, int D,
                         float* dfeat, mlsp_stream_t st) {
    return launch_interp3_bwd(st, dout, dist, rev_off, rev_ent, B, N, S, D, dfeat);
}
int mlsp_compose_linear_fwd_f32(const float* Wa, const float* ba, const float* Wb, const float* bb, int Cm, int Ci, int Co, float* W, float* b,
                                mlsp_stream_t st) {
    if (!Wa || !ba || !Wb || !bb || !W || !b || Cm <= 0 || Ci <= 0 || Co <= 0 || Cm > 1024 || Ci > 1024 || Co > 1024) return MLSP_ERR_ARG;
    return launch_compose_fwd((hipStream_t)st, Wa, ba, Wb, bb, Cm, Ci, Co, W, b);
}
int mlsp_compose_linear_bwd_f32(const float* dW, const float* db, const float* Wa, const float* ba, const float* Wb, int Cm, int Ci, int Co,
                                float* dWa, float* dba, float* dWb, mlsp_stream_t st) {
    if (!dW || !db || !Wa || !ba || !Wb || !dWa || !dba || !dWb || Cm <= 0 || Ci <= 0 || Co <= 0 || Cm > 1024 || Ci > 1024 || Co > 1024) return MLSP_ERR_ARG;
    return launch_compose_bwd((hipStream_t)st, dW, db, Wa, ba, Wb, Cm, Ci, Co, dWa, dba, dWb);
}

int mlsp_transform3_fwd_f32(const float* x, const float* T, int B, int N, float* out, mlsp_stream_t st) {
    return launch_transform3_fwd(st, x, T, B, N, out);
}
int mlsp_transform3_bwd_f32(const float* x, const float* T, const float* dout, int B, int N, float* dx, float* dT, mlsp_stream_t st) {
    return launch_transform3_bwd(st, x, T, dout, B, N, dx, dT);
}
int mlsp_collapse_to_point_f32(float* X, int B, int N, const int32_t* choice, const float* u, const float* noise, float radius2,
                               int min_pts, float* mask, int32_t* chosen, mlsp_stream_t st) {
    return launch_collapse_to_point(st, X, B, N, choice, u, noise, radius2, min_pts, mask, chosen);
}
int mlsp_scan_select_f32(const float* X, int B, int N, int C, const double* R, int pixel, float* Xs, float* mask, mlsp_stream_t st) {
    return launch_scan_select(st, X, B, N, C, R, pixel, Xs, mask);
}
int mlsp_fps_f32(const float* xyz, int ldx, int B, int N, int S, const int32_t* start, int32_t* fps_idx, mlsp_stream_t st) {
    return launch_fps(st, xyz, ldx, B, N, S, start, fps_idx);
}
int mlsp_ball_query_f32(const float* xyz, int ldx, const float* new_xyz, int ldq, int B, int N, int S, float radius_sq, int nsample,
                        int32_t* idx, mlsp_stream_t st) {
    return launch_ball_query(st, xyz, ldx, new_xyz, ldq, B, N, S, radius_sq, nsample, idx);
}
int mlsp_group_reverse(const int32_t* idx, int B, int S, int N, int ns, int32_t* rev_off, int32_t* rev_ent, mlsp_stream_t st) {
    return launch_group_reverse(st, idx, B, S, N, ns, rev_off, rev_ent);
}
int mlsp_sa_group_fwd_f32(const float* xyz, int ldx, const float* feat, int D, const float* new_xyz, int ldq, const int32_t* idx, int B,
                          int N, int S, int ns, float* G, mlsp_stream_t st) {
    return launch_sa_group_fwd(st, xyz, ldx, feat, D, new_xyz, ldq, idx, B, N, S, ns, G);
}
int mlsp_sa_group_bwd_f32(const float* dG, int ldg, int col, int D, const int32_t* rev_off, const int32_t* rev_ent, int B, int N, int S, int ns,
                          float* dfeat, mlsp_stream_t st) {
    return launch_sa_group_bwd(st, dG, ldg, col, D, rev_off, rev_ent, B, N, S, ns, dfeat);
}


int mlsp_sa_fold_fwd_f32(const float* u, const float* w, const int32_t* idx, int B, int N, int S, int ns, int C, const float* gamma,
                         const float* beta, float* run_mean, float* run_var, float momentum, float eps, int training, float* Z,
                         float* bn_save, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    if (B <= 0 || S <= 0 || ns <= 0 || C <= 0) return MLSP_ERR_ARG;
    Workspace wk(ws, ws_bytes);
    double* part = training ? wk.take<double>((size_t)sa_fold_parts((long)B * S * ns) * 2 * C) : nullptr;
    if (!wk.ok()) return MLSP_ERR_WORKSPACE;
    return launch_sa_fold_fwd(st, u, w, idx, B, N, S, ns, C, gamma, beta, run_mean, run_var, momentum, eps, training, Z, bn_save, part);
}

int mlsp_sa_fold_bwd_f32(const float* dZ, const float* u, const float* w, const int32_t* idx, const int32_t* rev_off, const int32_t* rev_ent,
                         const int32_t* rev_cnt, const int32_t* pad_cnt, int B, int N, int S, int ns, int C, const float* bn_save, int training,
                         float* du, float* dw, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    if (B <= 0 || S <= 0 || ns <= 0 || C <= 0) return MLSP_ERR_ARG;
    Workspace wk(ws, ws_bytes);
    double* part = wk.take<double>((size_t)sa_fold_parts((long)B * S * ns) * 2 * C);
    float* mean_dz = wk.take<float>(C);
    float* mean_dzy = wk.take<float>(C);
    if (!wk.ok()) return MLSP_ERR_WORKSPACE;
    return launch_sa_fold_bwd(st, dZ, u, w, idx, rev_off, rev_ent, B, N, S, ns, C, bn_save, training, du, dw, dgamma, dbeta, part, mean_dz,
                              mean_dzy, rev_cnt, pad_cnt);
}

int mlsp_group_reverse_compact(const int32_t* idx, int B, int S, int N, int ns, int32_t* rev_off, int32_t* rev_cnt, int32_t* rev_ent,
                               int32_t* pad_cnt, mlsp_stream_t st) {
    return launch_group_reverse_compact(st, idx, B, S, N, ns, rev_off, rev_cnt, rev_ent, pad_cnt);
}

int mlsp_graph_feature_fwd_f32(const float* x, const int32_t* idx, int B, int N, int C, int k, float* F, mlsp_stream_t st) {
    if (!x || !idx || !F || B <= 0 || N <= 0 || C <= 0 || k <= 0) return MLSP_ERR_ARG;
    return launch_graph_feature_fwd(st, x, idx, B * N, N, C, k, F);
}
int mlsp_graph_feature_bwd_f32(const float* dF, const int32_t* rev_off, const int32_t* rev_ent, int B, int N, int C, int k,
                               float* dx, mlsp_stream_t st) {
    if (!dF || !rev_off || !rev_ent || !dx || B <= 0 || N <= 0 || C <= 0 || k <= 0) return MLSP_ERR_ARG;
    return launch_graph_feature_bwd(st, dF, rev_off, rev_ent, B * N, N, C, k, dx);
}

int mlsp_edgeconv_fwd_f32(const float* x, int ldx, const int32_t* idx, const float* W, const float* gamma, const float* beta,
                          float* run_mean, float* run_var, float momentum, float eps, int act, float slope, int training,
                          int B, int N, int C, int Cout, int k, float* out, int ldo, float* uv, float* msel, uint8_t* argsel,
                          float* s1, float* bn_save, float* Wd_out, int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!x || !idx || !W || !gamma || !beta || !out || !uv || !msel || !argsel || !s1 || !bn_save) return MLSP_ERR_ARG;
    if (B <= 0 || N <= 0 || C <= 0 || Cout <= 0 || k <= 0 || k > 255 || ldx < C || ldo < Cout) return MLSP_ERR_ARG;
    const int P = B * N;
    Workspace w(ws, ws_bytes);
    float* Wd = Wd_out ? Wd_out : w.take<float>((size_t)2 * Cout * C);     // Wd_out: kept by the caller for the backward pass
    int nparts = edge_reduce_parts(P);
    double* part = w.take<double>((size_t)nparts * 2 * Cout);
    size_t sf = gemm_slab_floats(P, 2 * Cout, C);
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    float* scale = bn_save, *shift = bn_save + Cout, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
    if (!training) {                  // eval mode: the BatchNorm vectors do not depend on the batch -- prepared by the weight-fold launch
        if (!run_mean || !run_var) return MLSP_ERR_ARG;
        CHECK(launch_build_wd_eval(st, W, Cout, C, Wd, Cout, gamma, beta, run_mean, run_var, bn_save, 0, nullptr, nullptr, nullptr, nullptr, nullptr, eps));
    } else CHECK(launch_build_wd(st, W, Cout, C, Wd, (C >= 128 && C % 32 == 0 && build_wd_leaves_bound(Cout, C)) ? amax_reserve(Wd, 2 * Cout, C, C) : nullptr));   // (mode 3: the fold leaves the bound of what it writes)
    CHECK(launch_gemm(st, false, true, P, 2 * Cout, C, x, ldx, Wd, C, uv, 2 * Cout, nullptr, nullptr, 0, slab, sf));
    {   // the neighbour gather + max/min + BN sums: compulsory bytes = u half + indices in, msel + s1 + arg slot out
        const int tok = prof_cls_begin(st, MLSP_PROF_EDGE_REDUCE);
        const int rc = launch_edge_reduce(st, uv, idx, gamma, P, N, Cout, k, msel, argsel, s1, part, &nparts);
        prof_cls_end(st, tok, (double)P * (Cout * 4.0 + k * 4.0 + Cout * 9.0));
        if (rc != MLSP_OK) return rc;
    }
    if (training) {
        // (mode 3, the caller offered bounds for `out`: the finalizer leaves |gamma| sqrt(P k) + |beta| per channel -- a bound of the layer's
        // output for the GEMMs that read it, instead of a streaming pass over it)
        bn_bound_request(amax_offered_output(out, P, Cout, ldo, Cout));
        CHECK(launch_bn_finalize(st, part, nparts, (double)P * k, Cout, gamma, beta, run_mean, run_var, momentum, eps, scale,
                                 shift, mean, invstd));
    }
    CHECK(launch_edge_select_act(st, msel, uv, P, Cout, scale, shift, act, slope, out, ldo));
    return MLSP_OK;
}

int mlsp_edgeconv_bwd_f32(const float* dOut, int lddo, const float* x, int ldx, const int32_t* rev_off, const int32_t* rev_ent,
                          const float* W, const float* out, int ldo, const float* uv, const float* msel, const uint8_t* argsel,
                          const float* s1, const float* bn_save, const float* Wd_in, int act, float slope, int training, int B, int N, int C,
                          int Cout, int k, float* dx, int lddx, int dx_accumulate, float* dW, float* dgamma, float* dbeta, int precision, void* ws,
                          size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!dOut || !x || !rev_off || !rev_ent || !W || !out || !uv || !msel || !argsel || !s1 || !bn_save || !dW || !dgamma ||
        !dbeta)
        return MLSP_ERR_ARG;
    if (B <= 0 || N <= 0 || C <= 0 || Cout <= 0 || k <= 0 || ldx < C || lddo < Cout || ldo < Cout || (dx && lddx < C)) return MLSP_ERR_ARG;
    const int P = B * N;
    Workspace w(ws, ws_bytes);
    float* Wd = Wd_in ? nullptr : w.take<float>((size_t)2 * Cout * C);
    float* dWd = w.take<float>((size_t)2 * Cout * C);
    float* gz = w.take<float>((size_t)P * Cout);
    float* duv = w.take<float>((size_t)P * 2 * Cout);
    int nparts = bn_parts_max(P);
    double* part = w.take<double>((size_t)nparts * 2 * Cout);
    float* mean_dz = w.take<float>(Cout);
    float* mean_dzy = w.take<float>(Cout);
    size_t sf1 = gemm_slab_floats(P, C, 2 * Cout), sf2 = gemm_slab_floats(2 * Cout, C, P);
    size_t sf = sf1 > sf2 ? sf1 : sf2;
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    const float* scale = bn_save, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
    // f16x3: the passes that write duv leave its bound (256 partial maxima in a slot of the workspace tail, found again by both products
    // below) instead of a measuring pass over the finished tensor
    // (only where those products can take the two-piece kernel at all: C a multiple of its 128-column tile -- conv4 of the DGCNN encoder)
    float* duv_amax = (C % 128 == 0 && edge_bwd_leaves_duv_bound(dOut, out, msel, uv, s1, argsel, Cout, scale, mean, invstd, gz, duv, lddo, ldo))
                          ? amax_reserve(duv, P, 2 * Cout, 2 * Cout) : nullptr;
    CHECK(launch_edge_bwd_reduce(st, dOut, out, msel, uv, P, Cout, mean, invstd, act, slope, part, lddo, ldo, duv_amax));
    CHECK(launch_bn_bwd_finalize(st, part, edge_bwd_reduce_parts(P, Cout, dOut, out, msel, uv, mean, invstd, lddo, ldo), (double)P * k, Cout,
                                 dgamma, dbeta, mean_dz, mean_dzy));
    const float* mdz = training ? mean_dz : nullptr;
    CHECK(launch_edge_bwd_point(st, dOut, out, uv, s1, P, Cout, k, scale, mean, invstd, mdz, mean_dzy, act, slope, gz, duv, lddo, ldo, duv_amax));
    CHECK(launch_edge_bwd_gather(st, gz, argsel, uv, rev_off, rev_ent, P, N, Cout, scale, mean, invstd, mdz, mean_dzy, duv, duv_amax));
    // (the forward's [Wa ; Wb - Wa] is reused when the caller kept it; beta = 1: dx is added into a slice of a wider gradient)
    const float* Wdc = Wd_in;
    if (!Wdc && dx) { CHECK(launch_build_wd(st, W, Cout, C, Wd)); Wdc = Wd; }
    if (dx) CHECK(launch_gemm(st, false, false, P, C, 2 * Cout, duv, 2 * Cout, Wdc, C, dx, lddx, nullptr, nullptr, 0, slab, sf, nullptr,
                              nullptr, nullptr, nullptr, dx_accumulate != 0));
    gemm_unfold_request(dW);          // a split-K launch sums its slabs straight into the reference layout (gemm.hip splitk_reduce_unfold_kernel)
    const int rcw = launch_gemm(st, true, false, 2 * Cout, C, P, duv, 2 * Cout, x, ldx, dWd, C, nullptr, nullptr, 0, slab, sf);
    const bool unfolded = gemm_unfold_take();
    if (rcw != MLSP_OK) return rcw;
    if (!unfolded) CHECK(launch_unbuild_wd(st, dWd, Cout, C, dW));
    return MLSP_OK;
}

int mlsp_tnet_edge_fwd_f32(const float* x, int ldx, const int32_t* idx, const float* W1, const float* gamma1, const float* beta1,
                           float* run_mean1, float* run_var1, const float* W2, const float* gamma2, const float* beta2,
                           float* run_mean2, float* run_var2, float momentum, float eps, float slope, int training, int B, int N,
                           int C, int C1, int C2, int k, float* out, float* uv, float* s1, float* bn1_save, float* zsel,
                           uint8_t* argsel, float* bn2_save, int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!x || !idx || !W1 || !gamma1 || !beta1 || !W2 || !gamma2 || !beta2 || !out || !uv || !s1 || !bn1_save || !zsel || !argsel ||
        !bn2_save)
        return MLSP_ERR_ARG;
    if (B <= 0 || N <= 0 || C <= 0 || k <= 0 || ldx < C) return MLSP_ERR_ARG;
    if (C1 != 64 || C2 != 128 || tnet_points_per_tile(k) <= 0) return MLSP_ERR_UNSUPPORTED;
    const int P = B * N;
    Workspace w(ws, ws_bytes);
    float* Wd = w.take<float>((size_t)2 * C1 * C);
    float* msel = w.take<float>((size_t)P * C1);
    uint8_t* arg1 = w.take<uint8_t>((size_t)P * C1);
    int np1 = edge_reduce_parts(P), np2 = tnet_fwd_parts(B, N, k);
    double* part = w.take<double>((size_t)(np1 > np2 ? np1 : np2) * 2 * C2);
    size_t sf = gemm_slab_floats(P, 2 * C1, C);
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    if (!training) {                  // eval mode: both BatchNorm stages' vectors ride in the weight-fold launch
        if (!run_mean1 || !run_var1 || !run_mean2 || !run_var2) return MLSP_ERR_ARG;
        CHECK(launch_build_wd_eval(st, W1, C1, C, Wd, C1, gamma1, beta1, run_mean1, run_var1, bn1_save, C2, gamma2, beta2, run_mean2, run_var2, bn2_save, eps));
    } else CHECK(launch_build_wd(st, W1, C1, C, Wd));
    CHECK(launch_gemm(st, false, true, P, 2 * C1, C, x, ldx, Wd, C, uv, 2 * C1, nullptr, nullptr, 0, slab, sf));
    CHECK(launch_edge_reduce(st, uv, idx, gamma1, P, N, C1, k, msel, arg1, s1, part, &np1));
    if (training) {
        CHECK(launch_bn_finalize(st, part, np1, (double)P * k, C1, gamma1, beta1, run_mean1, run_var1, momentum, eps, bn1_save,
                                 bn1_save + C1, bn1_save + 2 * C1, bn1_save + 3 * C1));
    }
    {
        const int tok = prof_cls_begin(st, MLSP_PROF_TNET_FWD);
        const int rc = launch_tnet_edge_fwd(st, uv, idx, bn1_save, W2, gamma2, P, N, k, slope, zsel, argsel, part);
        prof_cls_end(st, tok, 2.0 * P * (double)k * C1 * C2);                  // the per-edge 64 -> 128 contraction
        if (rc != MLSP_OK) return rc;
    }
    if (training) {
        bn_bound_request(amax_offered_output(out, P, C2, C2, C2));     // (mode 3: the analytic bound of the stage's output, as mlsp_edgeconv_fwd_f32)
        CHECK(launch_bn_finalize(st, part, np2, (double)P * k, C2, gamma2, beta2, run_mean2, run_var2, momentum, eps, bn2_save,
                                 bn2_save + C2, bn2_save + 2 * C2, bn2_save + 3 * C2));
    }
    CHECK(launch_tnet_out(st, zsel, bn2_save, P, slope, out));
    return MLSP_OK;
}

int mlsp_tnet_edge_bwd_f32(const float* dOut, const float* x, int ldx, const int32_t* idx, const int32_t* rev_off,
                           const int32_t* rev_ent, const float* W1, const float* W2, const float* out, const float* uv,
                           const float* s1, const float* bn1_save, const float* zsel, const uint8_t* argsel, const float* bn2_save,
                           float slope, int training, int B, int N, int C, int C1, int C2, int k, float* dx, float* dW1,
                           float* dgamma1, float* dbeta1, float* dW2, float* dgamma2, float* dbeta2, int precision, void* ws, size_t ws_bytes,
                           mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!dOut || !x || !idx || !W1 || !W2 || !out || !uv || !s1 || !bn1_save || !zsel || !argsel || !bn2_save ||
        !dW1 || !dgamma1 || !dbeta1 || !dW2 || !dgamma2 || !dbeta2)
        return MLSP_ERR_ARG;
    if ((!rev_off || !rev_ent) && (dx || C > 4)) return MLSP_ERR_ARG;      // only the moment path (no input gradient, C <= 4) works without it
    if (B <= 0 || N <= 0 || C <= 0 || k <= 0 || ldx < C) return MLSP_ERR_ARG;
    if (C1 != 64 || C2 != 128 || tnet_points_per_tile(k) <= 0) return MLSP_ERR_UNSUPPORTED;
    const int P = B * N;
    const size_t E = (size_t)P * k;
    const int ntiles = B * ((N + tnet_points_per_tile(k) - 1) / tnet_points_per_tile(k));
    const int nb = tnet_grid(ntiles);
    Workspace w(ws, ws_bytes);
    float* Wd = w.take<float>((size_t)2 * C1 * C);
    float* dWd = w.take<float>((size_t)2 * C1 * C);
    float* g = w.take<float>((size_t)P * C2);
    float* coef = w.take<float>(2 * C2);
    float* dhp = w.take<float>(E * C1);
    float* dW2part = w.take<float>(tnet_bwd_scratch_floats(ntiles));
    int npr = (P + 511) / 512;
    const int nprv = bn_vec_parts(P);                                  // rows of the vectorised BN2 partial pass
    const int nrows_part = (npr > nb ? npr : nb) > nprv ? (npr > nb ? npr : nb) : nprv;
    double* part = w.take<double>((size_t)nrows_part * 2 * C2);
    float* mean_dz = w.take<float>(C2);
    float* mean_dzy = w.take<float>(C2);
    float* m1 = w.take<float>(C1);
    float* m2 = w.take<float>(C1);
    float* duv = w.take<float>((size_t)P * 2 * C1);
    size_t sf1 = gemm_slab_floats(P, C, 2 * C1), sf2 = gemm_slab_floats(2 * C1, C, P);
    size_t sf = sf1 > sf2 ? sf1 : sf2;
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    // BN2 backward sums of dz = dOut * act'(out): the same computation as the first pass of the generic BN backward (out > 0 <=>
    // scale2*zsel + shift2 > 0), so the vectorised column-stationary kernel is used when the operands allow it
    {
        const int rc = launch_bn_act_bwd_partials_vec(st, dOut, zsel, P, C2, bn2_save, bn2_save + C2, bn2_save + 2 * C2, bn2_save + 3 * C2,
                                                      2, slope, part);
        if (rc == MLSP_OK) npr = nprv;
        else if (rc != MLSP_ERR_UNSUPPORTED) return rc;
        else CHECK(launch_tnet_bwd_reduce(st, dOut, out, zsel, bn2_save, P, slope, part));
    }
    CHECK(launch_bn_bwd_finalize(st, part, npr, (double)E, C2, dgamma2, dbeta2, mean_dz, mean_dzy));
    CHECK(launch_tnet_bwd_g(st, dOut, out, bn2_save, training ? mean_dz : nullptr, mean_dzy, P, slope, g, coef));
    int nparts = 0;
    {
        const int tok = prof_cls_begin(st, MLSP_PROF_TNET_BWD);
        const int rc = launch_tnet_edge_bwd(st, uv, idx, bn1_save, W2, bn2_save, g, argsel, coef, P, N, k, slope, dhp, dW2part, part, dW2, &nparts);
        prof_cls_end(st, tok, 4.0 * P * (double)k * C1 * C2);                  // dgrad + wgrad of that contraction
        if (rc != MLSP_OK) return rc;
    }
    CHECK(launch_bn_bwd_finalize(st, part, nparts, (double)E, C1, dgamma1, dbeta1, m1, m2));
    if (!dx && C <= 4) {
        // the input cloud needs no gradient (DGCNN's T-Net reads the raw coordinates): dW1 straight from one sequential pass over dh'
        // and the coordinate moments -- no fold onto the points, no [P,128]^T [P,C] GEMM (tnet.hip, tnet_bwd_tmom_kernel)
        double* mom = w.take<double>(tnet_w1_moment_doubles(P, k));
        if (!w.ok()) return MLSP_ERR_WORKSPACE;
        return launch_tnet_bwd_w1_moments(st, dhp, idx, rev_off, x, ldx, W1, bn1_save, training ? m1 : nullptr, m2, P, N, k, C, mom, dW1);
    }
    CHECK(launch_tnet_edge_bwd2(st, dhp, uv, s1, bn1_save, training ? m1 : nullptr, m2, rev_off, rev_ent, P, N, k, duv));
    CHECK(launch_build_wd(st, W1, C1, C, Wd));
    if (dx) CHECK(launch_gemm(st, false, false, P, C, 2 * C1, duv, 2 * C1, Wd, C, dx, C, nullptr, nullptr, 0, slab, sf));
    CHECK(launch_gemm(st, true, false, 2 * C1, C, P, duv, 2 * C1, x, ldx, dWd, C, nullptr, nullptr, 0, slab, sf));
    CHECK(launch_unbuild_wd(st, dWd, C1, C, dW1));
    return MLSP_OK;
}

// The input of a chained layer (mlsp_defer_t): X points into the PRE-BatchNorm output of the previous layer; its BN scale / shift,
// activation and dropout are applied while the GEMM stages the operand (GemmXf), or by one streaming pass into the workspace when the shape
// is not covered.
static bool defer_ok(const mlsp_defer_t* in) {
    return in->bn_save && in->ld > 0 && in->col >= 0 && in->p_drop >= 0.f && in->p_drop < 1.f && in->act >= 0 && in->act <= 2;
}
static GemmXf chain_xf(const mlsp_defer_t& in, int which, int batch_stats) {
    // batch_stats: the producer's bn_save rows 2 / 3 are the BATCH mean / invstd of the matrix X points into (training mode)
    GemmXf x;
    x.scale = in.bn_save + in.col; x.shift = in.bn_save + in.ld + in.col; x.act = in.act; x.slope = in.slope; x.thresh = dropout_thresh8(in.p_drop);
    x.inv_keep = dropout_inv_keep8(in.p_drop); x.seed = in.seed; x.ld = in.ld; x.col = in.col; x.which = which;
    if (batch_stats) { x.mean = in.bn_save + 2 * in.ld + in.col; x.invstd = in.bn_save + 3 * in.ld + in.col; }
    return x;
}
// a LeakyReLU slope outside [0, 1] takes the streaming pass: the fused transform writes the activation as one max
static bool defer_fusable(const mlsp_defer_t& in) { return !(in.act == 2 && !(in.slope >= 0.f && in.slope <= 1.f)); }

static int pointmlp_fwd_impl(const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                             const float* gbias, int rows_per_group, const float* gamma, const float* beta, float* run_mean,
                             float* run_var, float momentum, float eps, int training, int act, float slope, float p_drop,
                             uint64_t seed, float* Y, float* Z, float* bn_save, void* ws, size_t ws_bytes, mlsp_stream_t st,
                             const mlsp_defer_t* in, float* group_ysum) {
    // group_ysum (nullable, [M / rows_per_group][Cout]; needs gbias + BatchNorm): the per-cloud column sums of Y, a by-product of the
    // statistics (one streaming pass when this shape's GEMM does not leave row-panel sums) -- what the backward needs for the per-cloud
    // bias gradient when it never forms dY (mlsp_pointmlp_bwd_*: group_ysum)
    if (!X || !W || (!Z && !gamma) || M <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldw < Cin) return MLSP_ERR_ARG;
    if (group_ysum && (!gamma || !gbias || rows_per_group <= 0 || M % rows_per_group || M <= 32)) return MLSP_ERR_ARG;
    if (gamma && (!beta || !Y || !bn_save)) return MLSP_ERR_ARG;
    if (p_drop < 0.f || p_drop >= 1.f) return MLSP_ERR_ARG;
    if (in && (!defer_ok(in) || in->col + Cin > in->ld)) return MLSP_ERR_ARG;
    // per-cloud layers (batch <= 32 rows): Linear + BatchNorm1d + activation + dropout in ONE kernel (skinny.hip)
    if (gamma && !gbias && M <= 32 && !in && Z)
        return launch_skinny_linear_bn_act(st, X, ldx, M, Cin, W, ldw, Cout, bias, gamma, beta, run_mean, run_var, momentum, eps,
                                           training, act, slope, p_drop, seed, Y, Z, bn_save);
    Workspace w(ws, ws_bytes);
    // BN batch statistics: fused into the GEMM epilogue (one partial per 128-row panel) unless the GEMM splits K
    const int fused_parts = (gamma && training) ? gemm_stat_parts(M, Cout, Cin) : 0;
    int nparts = fused_parts ? fused_parts : bn_stat_parts(M);
    double* part = gamma ? w.take<double>((size_t)nparts * 2 * Cout) : nullptr;
    size_t sf = gemm_slab_floats(M, Cout, Cin);
    float* slab = sf ? w.take<float>(sf) : nullptr;
    GemmXf xf_s; const GemmXf* xf = nullptr;
    if (in) {
        xf_s = chain_xf(*in, 1, training);
        if (defer_fusable(*in) && gemm_xf_supported(false, true, M, Cout, Cin, X, ldx, W, ldw, 1)) xf = &xf_s;
        else {                                             // shape outside the fused path: materialise the activated input once
            float* Xa = w.take<float>((size_t)M * Cin);
            if (!w.ok()) return MLSP_ERR_WORKSPACE;
            CHECK(launch_xf_materialize(st, X, ldx, M, Cin, xf_s, Xa));
            X = Xa; ldx = Cin;
        }
    }
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    if (!gamma) {
        // plain Linear: the GEMM writes Z directly.  Every activated layer of the hot path has a BN.
        if (act || p_drop > 0.f) return MLSP_ERR_UNSUPPORTED;
        return launch_gemm(st, false, true, M, Cout, Cin, X, ldx, W, ldw, Z, Cout, bias, gbias, rows_per_group, slab, sf, nullptr, nullptr,
                           nullptr, nullptr, false, xf);
    }
    float* scale = bn_save, *shift = bn_save + Cout, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
    CHECK(launch_gemm(st, false, true, M, Cout, Cin, X, ldx, W, ldw, Y, Cout, bias, gbias, rows_per_group, slab, sf,
                      fused_parts ? part : nullptr, nullptr, nullptr, nullptr, false, xf));
    const int prow = fused_parts ? M / fused_parts : 0;                    // rows per statistics panel
    const bool gsum_fin = group_ysum && training && fused_parts && M % fused_parts == 0 && rows_per_group % prow == 0;
    if (training) {
        if (!fused_parts) CHECK(launch_colstats(st, Y, M, Cout, Cout, part));
        if (gsum_fin) CHECK(launch_bn_finalize_groups(st, part, nparts, (double)M, Cout, gamma, beta, run_mean, run_var, momentum, eps, scale, shift,
                                                      mean, invstd, group_ysum, rows_per_group / prow));
        else CHECK(launch_bn_finalize(st, part, nparts, (double)M, Cout, gamma, beta, run_mean, run_var, momentum, eps, scale, shift,
                                      mean, invstd));
    } else {
        if (!run_mean || !run_var) return MLSP_ERR_ARG;
        CHECK(launch_bn_eval_prepare(st, Cout, gamma, beta, run_mean, run_var, eps, scale, shift, mean, invstd));
    }
    if (group_ysum && !gsum_fin) {
        float* gscr = w.take<float>((size_t)(M / rows_per_group) * 16 * Cout);
        if (!w.ok()) return MLSP_ERR_WORKSPACE;
        CHECK(launch_colsum_groups(st, Y, M / rows_per_group, rows_per_group, Cout, group_ysum, gscr));
    }
    // Z == NULL: the activation is deferred to the consumer (mlsp_pointmlp_*_chain_f32 apply it in their operand loads)
    if (Z) CHECK(launch_bn_act_fwd(st, Y, Z, (size_t)M, Cout, scale, shift, act, slope, training ? p_drop : 0.f, seed));
    return MLSP_OK;
}

int mlsp_pointmlp_fwd_f32(const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                          const float* gbias, int rows_per_group, const float* gamma, const float* beta, float* run_mean,
                          float* run_var, float momentum, float eps, int training, int act, float slope, float p_drop,
                          uint64_t seed, float* Y, float* Z, float* bn_save, float* group_ysum, int precision, void* ws, size_t ws_bytes,
                          mlsp_stream_t st) {
    PREC_SCOPE(precision);
    return pointmlp_fwd_impl(X, ldx, M, Cin, W, ldw, Cout, bias, gbias, rows_per_group, gamma, beta, run_mean, run_var, momentum, eps,
                             training, act, slope, p_drop, seed, Y, Z, bn_save, ws, ws_bytes, st, nullptr, group_ysum);
}

int mlsp_pointmlp_fwd_chain_f32(const float* Xpre, int ldx, const mlsp_defer_t* in, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                                const float* gbias, int rows_per_group, const float* gamma, const float* beta, float* run_mean,
                                float* run_var, float momentum, float eps, int training, int act, float slope, float p_drop,
                                uint64_t seed, float* Y, float* Z, float* bn_save, float* group_ysum, int precision, void* ws, size_t ws_bytes,
                                mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!in) return MLSP_ERR_ARG;
    return pointmlp_fwd_impl(Xpre, ldx, M, Cin, W, ldw, Cout, bias, gbias, rows_per_group, gamma, beta, run_mean, run_var, momentum, eps,
                             training, act, slope, p_drop, seed, Y, Z, bn_save, ws, ws_bytes, st, in, group_ysum);
}

static int pointmlp_bwd_impl(const float* dZ, const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* Y,
                             const float* bn_save, int has_bn, int training, int act, float slope, float p_drop, uint64_t seed,
                             int n_groups, int rows_per_group, float* dX, int lddx, int dx_accumulate, float* dW, float* dbias, float* dgbias,
                             float* dgamma, float* dbeta, void* ws, size_t ws_bytes, mlsp_stream_t st, const mlsp_defer_t* in,
                             double* in_stats, const double* pre_stats, int pre_parts, const float* group_ysum) {
    // in_stats (consumer role, needs `in`): dX is stored MASKED by the producer's activation derivative / dropout and the producer's
    // BatchNorm-backward column sums are left in in_stats [M / 128][2][in->ld] at column in->col (mlsp_pointmlp_bwd_stats_parts() > 0).
    // pre_stats (producer role): dZ arrives masked, its sums are in pre_stats [pre_parts][2][Cout]: no reduction pass.
    // pre_stats == NULL with pre_parts < 0: dZ arrives masked WITHOUT complete sums (only some consumers took part in this backward pass,
    // the other columns are zero): the reduction runs here and the mask is not applied a second time.
    if (!dZ || !X || !W || !dW || M <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldw < Cin) return MLSP_ERR_ARG;
    if (in_stats && (!in || !dX || dx_accumulate)) return MLSP_ERR_ARG;
    if (pre_stats && (!has_bn || pre_parts <= 0 || M <= 32)) return MLSP_ERR_ARG;
    if (!pre_stats && pre_parts < 0 && (!has_bn || M <= 32)) return MLSP_ERR_ARG;
    if (has_bn && (!Y || !bn_save || !dgamma || !dbeta)) return MLSP_ERR_ARG;
    if (dgbias && (n_groups <= 0 || rows_per_group <= 0 || (long)n_groups * rows_per_group != M)) return MLSP_ERR_ARG;
    Workspace w(ws, ws_bytes);
    float* dY = has_bn ? w.take<float>((size_t)M * Cout) : nullptr;
    int nparts = bn_parts_max(M);
    double* part = (has_bn || dbias) ? w.take<double>((size_t)nparts * 2 * Cout) : nullptr;
    float* mean_dz = has_bn ? w.take<float>(Cout) : nullptr;
    float* mean_dzy = has_bn ? w.take<float>(Cout) : nullptr;
    size_t sf1 = dX ? gemm_slab_floats(M, Cin, Cout) : 0, sf2 = gemm_slab_floats(Cout, Cin, M);
    size_t sf = sf1 > sf2 ? sf1 : sf2;
    float* slab = sf ? w.take<float>(sf) : nullptr;
    float* gscratch = dgbias ? w.take<float>((size_t)n_groups * 16 * Cout) : nullptr;
    float* coef = has_bn ? w.take<float>((size_t)4 * Cout) : nullptr;      // c0 | nk2 | sc | max |d'| (bn.hip bn_bwd_finalize_coef_kernel)
    const float* Xorig = X; const int ldx_orig = ldx;   // the previous layer's pre-BN output (fused statistics read it as it is)
    GemmXf xf_s; const GemmXf* xf = nullptr;            // chained input: the wgrad reads the previous layer's pre-BN output
    if (in) {
        if (!defer_ok(in) || in->col + Cin > in->ld || M <= 32) return MLSP_ERR_ARG;
        xf_s = chain_xf(*in, 2, training);
        if (defer_fusable(*in) && gemm_xf_supported(true, false, Cout, Cin, M, dZ, Cout, X, ldx, 2)) xf = &xf_s;
        else {
            float* Xa = w.take<float>((size_t)M * Cin);
            if (!w.ok()) return MLSP_ERR_WORKSPACE;
            CHECK(launch_xf_materialize(st, X, ldx, M, Cin, xf_s, Xa));
            X = Xa; ldx = Cin;
        }
    }
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    const float* g = dZ;   // gradient wrt the linear output
    int g_slabs = 0;       // > 0: launch_bn_act_bwd already left the per-group partial sums of dY in gscratch
    // masked gradient + its sums in hand (pre_stats): when the dgrad and the weight gradient can form dY = (d' + y * nk2 + c0) * sc in their
    // A operand loads (gemm.hip gemm_split_kernel<.., DY>) the apply pass and the dY tensor are skipped; the per-cloud bias gradient comes
    // from the row-panel sums of d' and the clouds' column sums of y (bn.hip launch_bn_dy_gbias)
    static const bool dy_off = getenv("MLSP_BWD_DY_OFF") != nullptr;        // read-once A/B switch (tools/ab)
    GemmDy dy_s = {Y, coef, Cout, 1}; const GemmDy* dy = nullptr;          // (the finalizers below fill the fourth coefficient row)
    if (has_bn && M > 32 && pre_stats && training && !dy_off && (!in || xf) && M % pre_parts == 0 &&
        (!dX || gemm_dy_supported(false, false, M, Cin, Cout, dZ, Cout, W, ldw)) && gemm_dy_supported(true, false, Cout, Cin, M, dZ, Cout, X, ldx) &&
        (!dgbias || (rows_per_group >= 256 && rows_per_group % (M / pre_parts) == 0 && Cout % 4 == 0 && 256 % (Cout / 4) == 0 && Cout <= 1024)))
        dy = &dy_s;
    float* zb = (dbias && has_bn && training) ? dbias : nullptr;   // a bias in front of a batch-statistics BatchNorm: analytically zero gradient,
    bool dbias_zeroed = false;                                     // written by the layer's finalizer where it has one per channel (no memset launch)
    if (dy && dgbias && group_ysum) {          // (the forward kept the clouds' column sums of y: the per-cloud bias gradient rides in the finalizer)
        CHECK(launch_bn_bwd_finalize_coef_groups(st, pre_stats, pre_parts, (double)M, Cout, bn_save, dgamma, dbeta, coef, group_ysum,
                                                 rows_per_group / (M / pre_parts), rows_per_group, dgbias));
    } else if (dy) {
        CHECK(launch_bn_bwd_finalize_coef_z(st, pre_stats, pre_parts, (double)M, Cout, bn_save, dgamma, dbeta, coef, zb));
        dbias_zeroed = zb != nullptr;
        if (dgbias) CHECK(launch_bn_dy_gbias(st, Y, n_groups, rows_per_group, Cout, pre_stats, M / pre_parts, coef, gscratch, dgbias, group_ysum));
    } else if (has_bn && M <= 32) {
        CHECK(launch_skinny_bn_bwd_z(st, dZ, Y, dY, M, Cout, bn_save, training, act, slope, p_drop, seed, dgamma, dbeta, zb));
        dbias_zeroed = zb != nullptr;
        g = dY;
    } else if (has_bn) {
        const float* scale = bn_save, *shift = bn_save + Cout, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
        // (the per-cloud bias gradient -- column sums of dY per cloud -- comes out of the same pass when the shape allows)
        bn_zero_vec_request(zb);
        const int rcb = launch_bn_act_bwd(st, dZ, Y, dY, M, Cout, scale, shift, mean, invstd, training, act, slope,
                                          training ? p_drop : 0.f, seed, part, dgamma, dbeta, mean_dz, mean_dzy, dgbias ? gscratch : nullptr,
                                          rows_per_group, &g_slabs, pre_stats, pre_parts);
        dbias_zeroed = bn_zero_vec_take();
        if (rcb != MLSP_OK) return rcb;
        g = dY;
    }
    GemmBs bs_s; const GemmBs* bs = nullptr;
    if (in_stats) {
        bs_s = {Xorig, ldx_orig, in->bn_save + in->col, in->ld, in->act, in->slope, dropout_thresh8(in->p_drop), dropout_inv_keep8(in->p_drop), in->seed,
                in->ld, in->col, in_stats + in->col, in->ld};
        bs_s.amax = (float*)(in_stats + (size_t)(M / 128) * 2 * in->ld) + in->col;      // the maxima plane behind the [M / 128][2][ld] sums
        bs = &bs_s;
    }
    static const bool no_pair = getenv("MLSP_SKINNY_NO_PAIR") != nullptr;          // read-once A/B switch
    if (M <= 32 && dX && !bs && !dy && !xf && !no_pair) {
        // per-cloud layer (rows = batch): input gradient (added into dX under dx_accumulate: the x5 halves of the PointSegDA heads share one
        // gradient buffer) and weight gradient in one launch (skinny.hip)
        CHECK(launch_skinny_bwd_pair(st, g, Cout, W, ldw, X, ldx, dX, lddx, dW, M, Cin, Cout, dx_accumulate));
    } else {
    if (dX) CHECK(launch_gemm(st, false, false, M, Cin, Cout, g, Cout, W, ldw, dX, lddx, nullptr, nullptr, 0, slab, sf, nullptr, nullptr,
                              nullptr, nullptr, dx_accumulate != 0, nullptr, 0, nullptr, bs, dy));
    CHECK(launch_gemm(st, true, false, Cout, Cin, M, g, Cout, X, ldx, dW, Cin, nullptr, nullptr, 0, slab, sf, nullptr, nullptr, nullptr,
                      nullptr, false, xf, 0, nullptr, nullptr, dy));
    }
    if (dbias) {
        if (has_bn && training) {
            // a bias in front of a batch-stat BN has an analytically zero gradient (sum_rows dY == 0)
            if (!dbias_zeroed) {
                hipError_t e = hipMemsetAsync(dbias, 0, (size_t)Cout * sizeof(float), st);
                if (e != hipSuccess) return (int)e;
            }
        } else {
            CHECK(launch_colsum(st, g, M, Cout, part, dbias));
        }
    }
    if (dgbias && !dy) {
        if (g_slabs > 0) CHECK(launch_colsum_groups_fin(st, gscratch, n_groups, Cout, g_slabs, dgbias));
        else CHECK(launch_colsum_groups(st, g, n_groups, rows_per_group, Cout, dgbias, gscratch));
    }
    return MLSP_OK;
}

int mlsp_pointmlp_bwd_f32(const float* dZ, const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* Y,
                          const float* bn_save, int has_bn, int training, int act, float slope, float p_drop, uint64_t seed,
                          int n_groups, int rows_per_group, float* dX, int lddx, int dx_accumulate, float* dW, float* dbias, float* dgbias,
                          float* dgamma, float* dbeta, const double* pre_stats, int pre_parts, const float* group_ysum, int precision, void* ws,
                          size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    return pointmlp_bwd_impl(dZ, X, ldx, M, Cin, W, ldw, Cout, Y, bn_save, has_bn, training, act, slope, p_drop, seed, n_groups,
                             rows_per_group, dX, lddx, dx_accumulate, dW, dbias, dgbias, dgamma, dbeta, ws, ws_bytes, st, nullptr, nullptr,
                             pre_stats, pre_parts, group_ysum);
}

// Row panels the fused statistics pass of mlsp_pointmlp_bwd_chain_f32(in_stats != NULL) writes for this layer shape (its dgrad
// dX [M][Cin] = dY [M][Cout] W): M / 128 when the dgrad runs on a kernel with that pass, else 0 (pass in_stats = NULL then).
int mlsp_pointmlp_bwd_stats_parts(int M, int Cin, int Cout, int ldw, int lddx, int precision) {
    if (precision < 0 || precision > 3) return 0;
    GemmPrecisionScope prec_scope_(precision);
    const int t = thin_bs_parts(M, Cin, Cout);
    return t ? t : gemm_bs_parts(M, Cin, Cout, Cout, ldw, lddx);
}

int mlsp_pointmlp_bwd_chain_f32(const float* dZ, const float* Xpre, int ldx, const mlsp_defer_t* in, int M, int Cin, const float* W, int ldw, int Cout, const float* Y,
                                const float* bn_save, int has_bn, int training, int act, float slope, float p_drop, uint64_t seed,
                                int n_groups, int rows_per_group, float* dX, int lddx, int dx_accumulate, float* dW, float* dbias,
                                float* dgbias, float* dgamma, float* dbeta, double* in_stats, const double* pre_stats, int pre_parts,
                                const float* group_ysum, int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!in) return MLSP_ERR_ARG;
    return pointmlp_bwd_impl(dZ, Xpre, ldx, M, Cin, W, ldw, Cout, Y, bn_save, has_bn, training, act, slope, p_drop, seed, n_groups,
                             rows_per_group, dX, lddx, dx_accumulate, dW, dbias, dgbias, dgamma, dbeta, ws, ws_bytes, st, in, in_stats,
                             pre_stats, pre_parts, group_ysum);
}

// ---- Linear + BatchNorm + act + max over the k rows of every group (last conv of a set-abstraction MLP + the neighbourhood max) ------
// The activated [M, Cout] tensor and its gradient are never materialised: forward keeps the pre-BN output Y (needed by the backward
// anyway), the group extreme ysel [G, Cout] and its slot; backward gets the BatchNorm sums from the G x Cout selected values (the
// gradient of the max is non-zero at one row per group and channel) and writes dY in one pass.
int mlsp_pointmlp_segmax_fwd_f32(const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                                 const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                 int training, int act, float slope, int k, float* Y, float* out, float* ysel, uint8_t* argk, float* bn_save,
                                 int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!X || !W || !gamma || !beta || !Y || !out || !ysel || !argk || !bn_save || M <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldw < Cin)
        return MLSP_ERR_ARG;
    if (k < 1 || k > 255 || M % k || Cout % 4) return MLSP_ERR_UNSUPPORTED;
    Workspace w(ws, ws_bytes);
    const int fused_parts = training ? gemm_stat_parts(M, Cout, Cin) : 0;
    int nparts = fused_parts ? fused_parts : bn_stat_parts(M);
    double* part = w.take<double>((size_t)nparts * 2 * Cout);
    size_t sf = gemm_slab_floats(M, Cout, Cin);
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    float* scale = bn_save, *shift = bn_save + Cout, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
    CHECK(launch_gemm(st, false, true, M, Cout, Cin, X, ldx, W, ldw, Y, Cout, bias, nullptr, 0, slab, sf, fused_parts ? part : nullptr));
    if (training) {
        if (!fused_parts) CHECK(launch_colstats(st, Y, M, Cout, Cout, part));
        CHECK(launch_bn_finalize(st, part, nparts, (double)M, Cout, gamma, beta, run_mean, run_var, momentum, eps, scale, shift, mean, invstd));
    } else {
        if (!run_mean || !run_var) return MLSP_ERR_ARG;
        CHECK(launch_bn_eval_prepare(st, Cout, gamma, beta, run_mean, run_var, eps, scale, shift, mean, invstd));
    }
    return launch_segsel_act_fwd(st, Y, M / k, k, Cout, scale, shift, act, slope, out, ysel, argk);
}

int mlsp_pointmlp_segmax_bwd_f32(const float* dOut, const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* Y,
                                 const float* ysel, const uint8_t* argk, const float* bn_save, int training, int act, float slope, int k,
                                 float* dX, int lddx, float* dW, float* dbias, float* dgamma, float* dbeta, int precision, void* ws, size_t ws_bytes,
                                 mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!dOut || !X || !W || !Y || !ysel || !argk || !bn_save || !dW || !dgamma || !dbeta || M <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin ||
        ldw < Cin) return MLSP_ERR_ARG;
    if (k < 1 || k > 255 || M % k || Cout % 4) return MLSP_ERR_UNSUPPORTED;
    const int G = M / k;
    Workspace w(ws, ws_bytes);
    float* dY = w.take<float>((size_t)M * Cout);
    const int npr = bn_vec_parts(G);
    double* part = w.take<double>((size_t)npr * 2 * Cout);
    float* mean_dz = w.take<float>(Cout);
    float* mean_dzy = w.take<float>(Cout);
    size_t sf1 = dX ? gemm_slab_floats(M, Cin, Cout) : 0, sf2 = gemm_slab_floats(Cout, Cin, M);
    size_t sf = sf1 > sf2 ? sf1 : sf2;
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    const float* scale = bn_save, *shift = bn_save + Cout, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
    // sums of dz' and dz'*yhat over ALL M rows == over the G x Cout selected entries (every other row's dz is zero)
    {
        const int rc = launch_bn_act_bwd_partials_vec(st, dOut, ysel, G, Cout, scale, shift, mean, invstd, act, slope, part);
        if (rc != MLSP_OK) return rc;                          // (vectorised shapes only: Cout % 4 == 0, 256 % (Cout / 4) == 0)
    }
    CHECK(launch_bn_bwd_finalize(st, part, npr, (double)M, Cout, dgamma, dbeta, mean_dz, mean_dzy));
    CHECK(launch_segsel_bwd_apply(st, dOut, Y, ysel, argk, (size_t)M, k, Cout, bn_save, training ? mean_dz : nullptr, mean_dzy, act, slope, dY));
    if (dX) CHECK(launch_gemm(st, false, false, M, Cin, Cout, dY, Cout, W, ldw, dX, lddx, nullptr, nullptr, 0, slab, sf));
    CHECK(launch_gemm(st, true, false, Cout, Cin, M, dY, Cout, X, ldx, dW, Cin, nullptr, nullptr, 0, slab, sf));
    if (dbias) {
        if (training) {                                        // a bias in front of a batch-statistics BatchNorm: analytically zero gradient
            hipError_t e = hipMemsetAsync(dbias, 0, (size_t)Cout * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        } else {
            CHECK(launch_colsum(st, dY, M, Cout, part, dbias));
        }
    }
    return MLSP_OK;
}

// ---- bf16 activation storage (BASELINE.json configs[4]) ---------------------------------------------------------------------------
// The Linear + BatchNorm + act (+dropout) layer with its activations held as bf16 in HBM: x_bf16 says how X (and dX) are stored,
// out_bf16 how Y, Z (and dZ, dY) are.  Weights, biases, BN parameters / statistics and every weight gradient stay fp32; products are
// bf16 x bf16 with fp32 accumulation (v_mfma_f32_32x32x16_bf16).  BN layers with fused statistics only (interior GEMM tiles):
// MLSP_ERR_UNSUPPORTED otherwise -- mlsp_pointmlp_mx_supported() tells the caller beforehand, which then keeps that layer in fp32.
int mlsp_pointmlp_mx_supported(int M, int Cin, int Cout, int ldx, int x_bf16, int training, int precision) {
    if (precision < 0 || precision > 3) return 0;
    GemmPrecisionScope prec_scope_(precision);
    if (M <= 32 || Cin % 32 || Cout % 128 || M % 128) return 0;
    if (x_bf16 ? (ldx % 8) : (ldx % 4)) return 0;
    if (Cout > 1024 || 256 % (Cout / 4)) return 0;
    if (training && gemm_stat_parts(M, Cout, Cin) <= 0) return 0;
    if (gemm_slab_floats(M, Cout, Cin) != 0 || gemm_slab_floats(M, Cin, Cout) != 0) return 0;      // forward / dgrad never split K here
    return 1;
}

int mlsp_pointmlp_fwd_mx(const void* X, int x_bf16, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                         const float* gbias, int rows_per_group, const float* gamma, const float* beta, float* run_mean,
                         float* run_var, float momentum, float eps, int training, int act, float slope, float p_drop, uint64_t seed,
                         void* Y, void* Z, int out_bf16, float* bn_save, int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!X || !W || !Y || !Z || !gamma || !beta || !bn_save || ldw < Cin || ldx < Cin) return MLSP_ERR_ARG;
    if (p_drop < 0.f || p_drop >= 1.f) return MLSP_ERR_ARG;
    if (!mlsp_pointmlp_mx_supported(M, Cin, Cout, ldx, x_bf16, training, precision)) return MLSP_ERR_UNSUPPORTED;
    Workspace w(ws, ws_bytes);
    const int fused_parts = training ? gemm_stat_parts(M, Cout, Cin) : 0;
    double* part = w.take<double>((size_t)(fused_parts ? fused_parts : 1) * 2 * Cout);
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    float* scale = bn_save, *shift = bn_save + Cout, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
    CHECK(launch_gemm_mx(st, false, true, M, Cout, Cin, X, x_bf16, ldx, W, 0, ldw, Y, out_bf16, Cout, bias, gbias, rows_per_group, nullptr, 0,
                         fused_parts ? part : nullptr, false));
    if (training) {
        CHECK(launch_bn_finalize(st, part, fused_parts, (double)M, Cout, gamma, beta, run_mean, run_var, momentum, eps, scale, shift,
                                 mean, invstd));
    } else {
        if (!run_mean || !run_var) return MLSP_ERR_ARG;
        CHECK(launch_bn_eval_prepare(st, Cout, gamma, beta, run_mean, run_var, eps, scale, shift, mean, invstd));
    }
    if (out_bf16) return launch_bn_act_fwd_b16(st, Y, Z, M, Cout, scale, shift, act, slope, training ? p_drop : 0.f, seed);
    return launch_bn_act_fwd(st, (const float*)Y, (float*)Z, (size_t)M, Cout, scale, shift, act, slope, training ? p_drop : 0.f, seed);
}

int mlsp_pointmlp_bwd_mx(const void* dZ, const void* X, int x_bf16, int ldx, int M, int Cin, const float* W, int ldw, int Cout,
                         const void* Y, int out_bf16, const float* bn_save, int training, int act, float slope, float p_drop,
                         uint64_t seed, int n_groups, int rows_per_group, void* dX, int lddx, int dx_accumulate, float* dW, float* dbias,
                         float* dgbias, float* dgamma, float* dbeta, int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!dZ || !X || !W || !Y || !bn_save || !dW || !dgamma || !dbeta || ldw < Cin || ldx < Cin) return MLSP_ERR_ARG;
    if (dgbias && (n_groups <= 0 || rows_per_group <= 0 || (long)n_groups * rows_per_group != M)) return MLSP_ERR_ARG;
    if (!mlsp_pointmlp_mx_supported(M, Cin, Cout, ldx, x_bf16, training, precision)) return MLSP_ERR_UNSUPPORTED;
    if (dbias && !training) return MLSP_ERR_UNSUPPORTED;
    Workspace w(ws, ws_bytes);
    void* dY = w.take<char>((size_t)M * Cout * (out_bf16 ? 2 : 4));
    const int nparts = bn_parts_max(M);
    double* part = w.take<double>((size_t)nparts * 2 * Cout);
    float* mean_dz = w.take<float>(Cout);
    float* mean_dzy = w.take<float>(Cout);
    const size_t sf = gemm_slab_floats(Cout, Cin, M);
    float* slab = sf ? w.take<float>(sf) : nullptr;
    float* gscratch = dgbias ? w.take<float>((size_t)n_groups * 16 * Cout) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    const float* scale = bn_save, *shift = bn_save + Cout, *mean = bn_save + 2 * Cout, *invstd = bn_save + 3 * Cout;
    bn_zero_vec_request(dbias);       // (a bias in front of a batch-stat BN: zero gradient, written by the finalizer of the pass below)
    const int rcb = out_bf16 ? launch_bn_act_bwd_b16(st, dZ, Y, dY, M, Cout, scale, shift, mean, invstd, training, act, slope, training ? p_drop : 0.f, seed,
                                                     part, dgamma, dbeta, mean_dz, mean_dzy)
                             : launch_bn_act_bwd(st, (const float*)dZ, (const float*)Y, (float*)dY, M, Cout, scale, shift, mean, invstd, training, act, slope,
                                                 training ? p_drop : 0.f, seed, part, dgamma, dbeta, mean_dz, mean_dzy);
    const bool dbias_zeroed = bn_zero_vec_take();
    if (rcb != MLSP_OK) return rcb;
    if (dX) CHECK(launch_gemm_mx(st, false, false, M, Cin, Cout, dY, out_bf16, Cout, W, 0, ldw, dX, x_bf16, lddx, nullptr, nullptr, 0, nullptr, 0,
                                 nullptr, dx_accumulate != 0));
    CHECK(launch_gemm_mx(st, true, false, Cout, Cin, M, dY, out_bf16, Cout, X, x_bf16, ldx, dW, 0, Cin, nullptr, nullptr, 0, slab, sf, nullptr, false));
    if (dbias && !dbias_zeroed) {     // a bias in front of a batch-stat BN has an analytically zero gradient (sum_rows dY == 0)
        hipError_t e = hipMemsetAsync(dbias, 0, (size_t)Cout * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if (dgbias) {
        if (out_bf16) CHECK(launch_colsum_groups_b16(st, dY, n_groups, rows_per_group, Cout, dgbias, gscratch));
        else CHECK(launch_colsum_groups(st, (const float*)dY, n_groups, rows_per_group, Cout, dgbias, gscratch));
    }
    return MLSP_OK;
}

int mlsp_pointmlp_colmax_fwd_f32(const float* X, int ldx, int B, int N, int Cin, const float* W, int ldw, int Cout,
                                 const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                 int training, int act, float slope, float* out, float* ysel, int32_t* arg, float* bn_save, int precision, void* ws,
                                 size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!X || !W || !gamma || !beta || !out || !ysel || !arg || !bn_save) return MLSP_ERR_ARG;
    if (B <= 0 || N <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldw < Cin) return MLSP_ERR_ARG;
    const int P = B * N;
    Workspace w(ws, ws_bytes);
    const int fused_parts = training ? gemm_stat_parts(P, Cout, Cin) : 0;
    // fully fused path: statistics AND the per-cloud column extreme come out of the GEMM epilogue, Y is never written
    const int prow = gemm_panel_rows(P, Cout, Cin);
    const bool fuse_sel = (N % prow == 0) && gemm_stat_parts(P, Cout, Cin) > 0;
    float* Y = fuse_sel ? nullptr : w.take<float>((size_t)P * Cout);
    const int ntm = (P + prow - 1) / prow;
    float* pv = fuse_sel ? w.take<float>((size_t)ntm * Cout) : nullptr;
    int* pr = fuse_sel ? w.take<int>((size_t)ntm * Cout) : nullptr;
    int nparts = fused_parts ? fused_parts : bn_stat_parts(P);
    double* part = w.take<double>((size_t)nparts * 2 * Cout);
    size_t sf = gemm_slab_floats(P, Cout, Cin);
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    CHECK(launch_gemm(st, false, true, P, Cout, Cin, X, ldx, W, ldw, Y, Cout, nullptr, nullptr, 0, slab, sf,
                      fused_parts ? part : nullptr, fuse_sel ? gamma : nullptr, pv, pr));
    if (training) {
        if (!fused_parts) CHECK(launch_colstats(st, Y, P, Cout, Cout, part));
        CHECK(launch_bn_finalize(st, part, nparts, (double)P, Cout, gamma, beta, run_mean, run_var, momentum, eps, bn_save,
                                 bn_save + Cout, bn_save + 2 * Cout, bn_save + 3 * Cout));
    } else {
        if (!run_mean || !run_var) return MLSP_ERR_ARG;
        CHECK(launch_bn_eval_prepare(st, Cout, gamma, beta, run_mean, run_var, eps, bn_save, bn_save + Cout, bn_save + 2 * Cout,
                                     bn_save + 3 * Cout));
    }
    if (fuse_sel) CHECK(launch_colsel_panels(st, pv, pr, gamma, B, N, Cout, prow, ysel, arg, bn_save, act, slope, out));   // selection + BN + activation: one pass
    else {
        CHECK(launch_colsel(st, Y, gamma, B, N, Cout, ysel, arg));
        CHECK(launch_colsel_out(st, ysel, bn_save, B, Cout, act, slope, out));
    }
    return MLSP_OK;
}

int mlsp_pointmlp_colmax_bwd_f32(const float* dOut, const float* X, int ldx, int B, int N, int Cin, const float* W, int ldw,
                                 int Cout, const float* out, const float* ysel, const int32_t* arg, const float* bn_save,
                                 int training, int act, float slope, float* dX, int dx_accumulate, float* dW, float* dgamma, float* dbeta,
                                 int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    if (!dOut || !X || !W || !out || !ysel || !arg || !bn_save || !dW || !dgamma || !dbeta) return MLSP_ERR_ARG;
    if (B <= 0 || N <= 0 || Cin <= 0 || Cout <= 0 || ldx != Cin || ldw < Cin) return MLSP_ERR_ARG;
    const int P = B * N;
    Workspace w(ws, ws_bytes);
    float* g = w.take<float>((size_t)B * Cout);
    float* coef = w.take<float>((size_t)4 * Cout);
    float* S = w.take<float>((size_t)Cout * Cin);
    float* sx = w.take<float>(Cin);
    double* part = w.take<double>((size_t)bn_parts_max(P) * 2 * Cin);
    float* G = w.take<float>((size_t)Cin * Cin);
    float* WG = w.take<float>((size_t)Cout * Cin);
    float* Wb = w.take<float>((size_t)Cout * Cin);
    float* Mneg = w.take<float>((size_t)Cin * Cin);
    float* negr = w.take<float>(Cin);
    size_t sf = gemm_slab_floats(Cin, Cin, P), s2 = gemm_slab_floats(Cout, Cin, Cin), s3 = gemm_slab_floats(Cin, Cin, Cout),
           s4 = gemm_slab_floats(P, Cin, Cin);
    if (s2 > sf) sf = s2;
    if (s3 > sf) sf = s3;
    if (s4 > sf) sf = s4;
    float* slab = sf ? w.take<float>(sf) : nullptr;
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    CHECK(launch_colmax_bwd_coef(st, dOut, out, ysel, bn_save, B, Cout, (double)P, act, slope, training, g, coef, dgamma, dbeta));
    CHECK(launch_colmax_gather_rows(st, g, arg, X, ldx, B, N, Cout, Cin, S));
    if (training) {
        CHECK(launch_colsum(st, X, P, Cin, part, sx));
        CHECK(launch_gemm(st, true, false, Cin, Cin, P, X, ldx, X, ldx, G, Cin, nullptr, nullptr, 0, slab, sf));
        CHECK(launch_gemm(st, false, false, Cout, Cin, Cin, W, ldw, G, Cin, WG, Cin, nullptr, nullptr, 0, slab, sf));
        CHECK(launch_colmax_dw(st, S, WG, sx, coef, bn_save, Cout, Cin, dW));
    } else {
        hipError_t e = hipMemcpyAsync(dW, S, (size_t)Cout * Cin * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    if (dX) {
        if (training) {
            CHECK(launch_wt_vec_neg_scale_rows(st, W, ldw, coef + 2 * Cout, coef + 3 * Cout, Cout, Cin, negr, Wb));    // negr; Wb = -Bc * W
            CHECK(launch_gemm(st, true, false, Cin, Cin, Cout, Wb, Cin, W, ldw, Mneg, Cin, nullptr, nullptr, 0, slab, sf));
            CHECK(launch_gemm(st, false, false, P, Cin, Cin, X, ldx, Mneg, Cin, dX, Cin, negr, nullptr, 0, slab, sf, nullptr, nullptr,
                              nullptr, nullptr, dx_accumulate != 0));
        } else if (!dx_accumulate) {
            hipError_t e = hipMemsetAsync(dX, 0, (size_t)P * Cin * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
        CHECK(launch_colmax_scatter_rows(st, g, arg, W, ldw, B, N, Cout, Cin, dX, Cin));
    }
    return MLSP_OK;
}

int mlsp_radius_count_f32(const float* x, int ldx, int B, int N, float radius, int max_nn, int32_t* count, mlsp_stream_t st) {
    if (!x || !count || B <= 0 || N <= 0 || ldx < 3 || !(radius > 0.f) || max_nn <= 0) return MLSP_ERR_ARG;
    return launch_radius_count(st, x, ldx, B, N, radius, max_nn, count);
}
int mlsp_knn_normals_f32(const float* x, int ldx, const int32_t* idx, int B, int N, int k, float* normals, mlsp_stream_t st) {
    if (!x || !idx || !normals || B <= 0 || N <= 0 || k < 3 || ldx < 3) return MLSP_ERR_ARG;
    return launch_knn_normals(st, x, ldx, idx, B, N, k, normals);
}

int mlsp_segmax_fwd_f32(const float* Z, int P, int k, int C, float* out, uint8_t* argk, mlsp_stream_t st) {
    if (!Z || !out || !argk || P <= 0 || k <= 0 || k > 255 || C <= 0) return MLSP_ERR_ARG;
    return launch_segmax_fwd(st, Z, P, k, C, out, argk);
}
int mlsp_segmax_bwd_f32(const float* dOut, const uint8_t* argk, int P, int k, int C, float* dZ, mlsp_stream_t st) {
    if (!dOut || !argk || !dZ || P <= 0 || k <= 0 || C <= 0) return MLSP_ERR_ARG;
    return launch_segmax_bwd(st, dOut, argk, P, k, C, dZ);
}
int mlsp_colmax_fwd_f32(const float* Z, int B, int N, int C, float* out, int32_t* arg, mlsp_stream_t st) {
    if (!Z || !out || !arg || B <= 0 || N <= 0 || C <= 0) return MLSP_ERR_ARG;
    return launch_colmax_fwd(st, Z, B, N, C, out, arg);
}
int mlsp_colmax_bwd_f32(const float* dOut, const int32_t* arg, int B, int N, int C, float* dZ, mlsp_stream_t st) {
    if (!dOut || !arg || !dZ || B <= 0 || N <= 0 || C <= 0) return MLSP_ERR_ARG;
    return launch_colmax_bwd(st, dOut, arg, B, N, C, dZ);
}

int mlsp_chamfer_masked_fwd_f32(const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                                float* per_cloud, int32_t* argA, int32_t* argB, float* loss, mlsp_stream_t st) {
    if (!pred || !gold || !mask || !per_cloud || !argA || !argB || !loss || B <= 0 || N <= 0) return MLSP_ERR_ARG;
    return launch_chamfer_fwd(st, pred, gold, mask, B, N, scale, per_cloud, argA, argB, loss);
}
int mlsp_chamfer_masked_bwd_f32(const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                                const float* per_cloud, const int32_t* argA, const int32_t* argB, const float* grad_loss,
                                float* dpred, mlsp_stream_t st) {
    if (!pred || !gold || !mask || !per_cloud || !argA || !argB || !grad_loss || !dpred || B <= 0 || N <= 0) return MLSP_ERR_ARG;
    return launch_chamfer_bwd(st, pred, gold, mask, B, N, scale, per_cloud, argA, argB, grad_loss, dpred);
}

int mlsp_chamfer_dir_fwd_f32(const float* p1, const float* p2, const float* mask_cord, int B, int N, float* per_cloud, int32_t* arg,
                             float* loss, mlsp_stream_t st) {
    if (!p1 || !p2 || !mask_cord || !per_cloud || !arg || !loss || B <= 0 || N <= 0) return MLSP_ERR_ARG;
    return launch_chamfer_dir_fwd(st, p1, p2, mask_cord, B, N, per_cloud, arg, loss);
}
int mlsp_chamfer_dir_bwd_f32(const float* p1, const float* p2, const float* mask_cord, int B, int N, const float* per_cloud,
                             const int32_t* arg, const float* grad_loss, float* dp1, float* dp2, mlsp_stream_t st) {
    if (!p1 || !p2 || !mask_cord || !per_cloud || !arg || !grad_loss || (!dp1 && !dp2) || B <= 0 || N <= 0) return MLSP_ERR_ARG;
    return launch_chamfer_dir_bwd(st, p1, p2, mask_cord, B, N, per_cloud, arg, grad_loss, dp1, dp2);
}

int mlsp_normal_loss_fwd_f32(const float* pred, const float* gt, const float* wgt, int P, float weight, float* out, void* ws,
                             size_t ws_bytes, mlsp_stream_t st) {
    if (!pred || !gt || !out || P <= 0) return MLSP_ERR_ARG;
    Workspace w(ws, ws_bytes);
    double* part = w.take<double>(256 * 2);
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    return launch_normal_loss_fwd(st, pred, gt, wgt, P, weight, part, out);
}
int mlsp_normal_loss_bwd_f32(const float* pred, const float* gt, const float* wgt, int P, float weight, const float* fwd_out,
                             const float* grad_loss, float* dpred, mlsp_stream_t st) {
    if (!pred || !gt || !fwd_out || !grad_loss || !dpred || P <= 0) return MLSP_ERR_ARG;
    return launch_normal_loss_bwd(st, pred, gt, wgt, P, weight, fwd_out, grad_loss, dpred);
}

int mlsp_density_tail_fwd_f32(const float* logits, const float* fc2w, int P, int nc, float* pvec, float* dens, mlsp_stream_t st) {
    if (!logits || !fc2w || !pvec || !dens || P <= 0 || nc <= 0) return MLSP_ERR_ARG;
    return launch_density_tail_fwd(st, logits, fc2w, P, nc, pvec, dens);
}
int mlsp_density_tail_bwd_f32(const float* pvec, const float* fc2w, const float* dpvec, const float* ddens, int P, int nc,
                              float* dlogits, mlsp_stream_t st) {
    if (!pvec || !fc2w || !dlogits || P <= 0 || nc <= 0) return MLSP_ERR_ARG;
    return launch_density_tail_bwd(st, pvec, fc2w, dpvec, ddens, P, nc, dlogits);
}

int mlsp_density_loss_fwd_f32(const float* pvec, const float* dens, const float* target_vec, const float* target,
                              const float* mask, int P, int nc, float density_weight, float* out, void* ws, size_t ws_bytes,
                              mlsp_stream_t st) {
    if (!pvec || !dens || !target_vec || !target || !out || P <= 0 || nc <= 0) return MLSP_ERR_ARG;
    Workspace w(ws, ws_bytes);
    double* part = w.take<double>(256 * 3);
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    return launch_density_loss_fwd(st, pvec, dens, target_vec, target, mask, P, nc, density_weight, part, out);
}
int mlsp_density_loss_bwd_f32(const float* pvec, const float* dens, const float* target_vec, const float* target,
                              const float* mask, int P, int nc, float density_weight, const float* fwd_out,
                              const float* grad_kl, const float* grad_mae, float* dpvec, float* ddens, mlsp_stream_t st) {
    if (!pvec || !dens || !target_vec || !target || !fwd_out || !dpvec || !ddens || P <= 0 || nc <= 0) return MLSP_ERR_ARG;
    return launch_density_loss_bwd(st, pvec, dens, target_vec, target, mask, P, nc, density_weight, fwd_out, grad_kl, grad_mae,
                                   dpvec, ddens);
}

}  // extern "C"
