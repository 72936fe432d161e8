// Scoring and ranking (MFRecommender.py:99-133): gather + dot kernels are hand
// written; the argsort is a stable descending radix sort (rocPRIM), which
// reproduces torch.argsort(descending=True)'s order including ties by position.
#include "common.h"

namespace daisy {

template <class C>
__global__ __launch_bounds__(kBlock) void k_predict(const float *__restrict__ P,
                                                    const float *__restrict__ Q, int d,
                                                    const int64_t *__restrict__ u,
                                                    const int64_t *__restrict__ i, int64_t B,
                                                    float *__restrict__ out,
                                                    const float *__restrict__ bu,
                                                    const float *__restrict__ bi,
                                                    const float *__restrict__ b0) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    for (int64_t s = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; s < B; s += gstride) {
        Row<C> p, q;
        p.load(P + u[s] * d, lane, d);
        q.load(Q + i[s] * d, lane, d);
        float x = row_dot<C>(p, q);
        if (bu) x += (bu[u[s]] + bi[i[s]]) + b0[0];        // FMRecommender.py:65-66
        if (lane == 0) out[s] = x;
    }
}

// scores[b][c] = <P[us[b]], Q[cands[b][c]]>   (the bmm of MFRecommender.py:113-115)
template <class C>
__global__ __launch_bounds__(kBlock) void k_rank_scores(const float *__restrict__ P,
                                                        const float *__restrict__ Q, int d,
                                                        const int64_t *__restrict__ us,
                                                        const int64_t *__restrict__ cands,
                                                        int64_t B, int64_t Cn,
                                                        float *__restrict__ scores,
                                                        const float *__restrict__ bu,
                                                        const float *__restrict__ bi,
                                                        const float *__restrict__ b0) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    const int64_t n = B * Cn;
    for (int64_t e = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; e < n; e += gstride) {
        const int64_t b = e / Cn;
        Row<C> p, q;
        p.load(P + us[b] * d, lane, d);
        q.load(Q + cands[e] * d, lane, d);
        float x = row_dot<C>(p, q);
        if (bu) x += (bu[us[b]] + bi[cands[e]]) + b0[0];   // FMRecommender.py:115
        if (lane == 0) scores[e] = x;
    }
}

// scores[r] = <P[u], Q[r]> for every item (MFRecommender.py:129-131)
template <class C>
__global__ __launch_bounds__(kBlock) void k_scores_all(const float *__restrict__ P,
                                                       const float *__restrict__ Q, int d, int64_t u,
                                                       int64_t I, float *__restrict__ scores,
                                                       int64_t *__restrict__ ids,
                                                       const float *__restrict__ bu,
                                                       const float *__restrict__ bi,
                                                       const float *__restrict__ b0) {
    const int lane = threadIdx.x % C::LPR;
    const int group = threadIdx.x / C::LPR;
    const int64_t gstride = (int64_t)gridDim.x * C::GROUPS_PER_BLOCK;
    Row<C> p;
    p.load(P + u * d, lane, d);
    for (int64_t r = (int64_t)blockIdx.x * C::GROUPS_PER_BLOCK + group; r < I; r += gstride) {
        Row<C> q;
        q.load(Q + r * d, lane, d);
        float x = row_dot<C>(p, q);
        if (bu) x += (bu[u] + bi[r]) + b0[0];              // FMRecommender.py:131
        if (lane == 0) {
            scores[r] = x;
            ids[r] = r;
        }
    }
}

__global__ void k_take_topk(const int64_t *__restrict__ sorted_ids, int64_t B, int64_t Cn, int topk,
                            int64_t *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < B * topk;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / topk, t = e % topk;
        out[e] = sorted_ids[b * Cn + t];
    }
}

static inline hipStream_t S(daisy_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__global__ void k_iota_i64(int64_t n, int64_t *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = e;
}

}  // namespace daisy

using namespace daisy;

extern "C" {

int daisy_mf_predict(const float *P, const float *Q, int32_t d, const int64_t *u, const int64_t *i,
                     int64_t B, float *out, daisy_stream_t stream) {
    return daisy_fm_predict(P, Q, nullptr, nullptr, nullptr, d, u, i, B, out, stream);
}

int daisy_fm_predict(const float *P, const float *Q, const float *u_bias, const float *i_bias,
                     const float *bias, int32_t d, const int64_t *u, const int64_t *i, int64_t B,
                     float *out, daisy_stream_t stream) {
    DAISY_CHECK_ARG(P && Q && u && i && out && B > 0, "mf_predict: bad argument");
    DAISY_CHECK_ARG(!u_bias || (i_bias && bias), "fm_predict: u_bias without i_bias / bias");
    int rc = dispatch_d(d, [&](auto cfg) {
        using C = decltype(cfg);
        hipLaunchKernelGGL((k_predict<C>), dim3(grid_for(B, C::GROUPS_PER_BLOCK)), dim3(kBlock), 0,
                           S(stream), P, Q, (int)d, u, i, B, out, u_bias, i_bias, bias);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

size_t daisy_mf_rank_workspace_bytes(int64_t B, int64_t C) {
    if (B <= 0 || C <= 0) return 0;
    const size_t n = (size_t)B * (size_t)C;
    return align_up(n * 4) * 2 + align_up(n * 8) + align_up(seg_sort_desc_f32_i64_temp_bytes(n, B));
}

int daisy_mf_rank_topk(const float *P, const float *Q, int32_t d, const int64_t *us,
                       const int64_t *cands, int64_t B, int64_t C, int32_t topk, int64_t *out_ids,
                       float *scores_out, void *workspace, size_t workspace_bytes,
                       daisy_stream_t stream) {
    return daisy_fm_rank_topk(P, Q, nullptr, nullptr, nullptr, d, us, cands, B, C, topk, out_ids,
                              scores_out, workspace, workspace_bytes, stream);
}

int daisy_fm_rank_topk(const float *P, const float *Q, const float *u_bias, const float *i_bias,
                       const float *bias, int32_t d, const int64_t *us, const int64_t *cands,
                       int64_t B, int64_t C, int32_t topk, int64_t *out_ids, float *scores_out,
                       void *workspace, size_t workspace_bytes, daisy_stream_t stream) {
    DAISY_CHECK_ARG(!u_bias || (i_bias && bias), "fm_rank_topk: u_bias without i_bias / bias");
    DAISY_CHECK_ARG(P && Q && us && cands && out_ids && workspace, "mf_rank_topk: NULL argument");
    DAISY_CHECK_ARG(B > 0 && C > 0 && topk > 0 && topk <= C && B * C < ((int64_t)1 << 31),
                    "mf_rank_topk: bad sizes B=%lld C=%lld topk=%d", (long long)B, (long long)C, topk);
    DAISY_CHECK_ARG(workspace_bytes >= daisy_mf_rank_workspace_bytes(B, C),
                    "mf_rank_topk: workspace too small");
    hipStream_t s = S(stream);
    const size_t n = (size_t)B * (size_t)C;
    char *w = (char *)workspace;
    float *scores = (float *)w;             w += align_up(n * 4);
    float *sorted_scores = (float *)w;      w += align_up(n * 4);
    int64_t *sorted_ids = (int64_t *)w;     w += align_up(n * 8);
    void *temp = w;
    const size_t temp_bytes = seg_sort_desc_f32_i64_temp_bytes(n, B);
    int rc = dispatch_d(d, [&](auto cfg) {
        using Cf = decltype(cfg);
        hipLaunchKernelGGL((k_rank_scores<Cf>), dim3(grid_for(B * C, Cf::GROUPS_PER_BLOCK * 4)),
                           dim3(kBlock), 0, s, P, Q, (int)d, us, cands, B, C, scores, u_bias, i_bias, bias);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    if (scores_out) DAISY_HIP(hipMemcpyAsync(scores_out, scores, n * 4, hipMemcpyDeviceToDevice, s));
    rc = seg_sort_desc_f32_i64(temp, temp_bytes, scores, sorted_scores, cands, sorted_ids, B, C, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_take_topk, dim3(grid_for(B * topk, kBlock)), dim3(kBlock), 0, s, sorted_ids, B,
                       C, (int)topk, out_ids);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

size_t daisy_mf_full_rank_workspace_bytes(int64_t item_num) {
    if (item_num <= 0) return 0;
    const size_t n = (size_t)item_num;
    return align_up(n * 4) * 2 + align_up(n * 8) * 2 + align_up(sort_pairs_desc_f32_i64_temp_bytes(n));
}

int daisy_mf_full_rank(const float *P, const float *Q, int32_t d, int64_t item_num, int64_t u,
                       int32_t topk, int64_t *out_ids, void *workspace, size_t workspace_bytes,
                       daisy_stream_t stream) {
    return daisy_fm_full_rank(P, Q, nullptr, nullptr, nullptr, d, item_num, u, topk, out_ids, workspace,
                              workspace_bytes, stream);
}

int daisy_fm_full_rank(const float *P, const float *Q, const float *u_bias, const float *i_bias,
                       const float *bias, int32_t d, int64_t item_num, int64_t u, int32_t topk,
                       int64_t *out_ids, void *workspace, size_t workspace_bytes,
                       daisy_stream_t stream) {
    DAISY_CHECK_ARG(!u_bias || (i_bias && bias), "fm_full_rank: u_bias without i_bias / bias");
    DAISY_CHECK_ARG(P && Q && out_ids && workspace && item_num > 0 && u >= 0 && topk > 0 &&
                        topk <= item_num,
                    "mf_full_rank: bad argument");
    DAISY_CHECK_ARG(workspace_bytes >= daisy_mf_full_rank_workspace_bytes(item_num),
                    "mf_full_rank: workspace too small");
    hipStream_t s = S(stream);
    const size_t n = (size_t)item_num;
    char *w = (char *)workspace;
    float *scores = (float *)w;          w += align_up(n * 4);
    float *sorted_scores = (float *)w;   w += align_up(n * 4);
    int64_t *ids = (int64_t *)w;         w += align_up(n * 8);
    int64_t *sorted_ids = (int64_t *)w;  w += align_up(n * 8);
    void *temp = w;
    int rc = dispatch_d(d, [&](auto cfg) {
        using Cf = decltype(cfg);
        hipLaunchKernelGGL((k_scores_all<Cf>), dim3(grid_for(item_num, Cf::GROUPS_PER_BLOCK * 4)),
                           dim3(kBlock), 0, s, P, Q, (int)d, u, item_num, scores, ids, u_bias, i_bias, bias);
        return DAISY_OK;
    });
    if (rc) return rc;
    DAISY_LAUNCH_CHECK();
    rc = sort_pairs_desc_f32_i64(temp, sort_pairs_desc_f32_i64_temp_bytes(n), scores, sorted_scores,
                                 ids, sorted_ids, item_num, s);
    if (rc) return rc;
    DAISY_HIP(hipMemcpyAsync(out_ids, sorted_ids, (size_t)topk * 8, hipMemcpyDeviceToDevice, s));
    return DAISY_OK;
}

int daisy_topk_from_scores(const float *scores, const int64_t *cands, int64_t B, int64_t C, int32_t topk,
                           int64_t *out_ids, void *workspace, size_t workspace_bytes,
                           daisy_stream_t stream) {
    DAISY_CHECK_ARG(scores && cands && out_ids && workspace, "topk_from_scores: NULL argument");
    DAISY_CHECK_ARG(B > 0 && C > 0 && topk > 0 && topk <= C && B * C < ((int64_t)1 << 31),
                    "topk_from_scores: bad sizes B=%lld C=%lld topk=%d", (long long)B, (long long)C, topk);
    DAISY_CHECK_ARG(workspace_bytes >= daisy_mf_rank_workspace_bytes(B, C), "topk_from_scores: workspace too small");
    hipStream_t s = S(stream);
    const size_t n = (size_t)B * (size_t)C;
    char *w = (char *)workspace;
    w += align_up(n * 4);                                 // (the slot the MF path keeps its scores in)
    float *sorted_scores = (float *)w;      w += align_up(n * 4);
    int64_t *sorted_ids = (int64_t *)w;     w += align_up(n * 8);
    int rc = seg_sort_desc_f32_i64(w, seg_sort_desc_f32_i64_temp_bytes(n, B), scores, sorted_scores, cands,
                                   sorted_ids, B, C, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_take_topk, dim3(grid_for(B * topk, kBlock)), dim3(kBlock), 0, s, sorted_ids, B, C,
                       (int)topk, out_ids);
    DAISY_LAUNCH_CHECK();
    return DAISY_OK;
}

int daisy_full_topk_from_scores(const float *scores, int64_t item_num, int32_t topk, int64_t *out_ids,
                                void *workspace, size_t workspace_bytes, daisy_stream_t stream) {
    DAISY_CHECK_ARG(scores && out_ids && workspace && item_num > 0 && topk > 0 && topk <= item_num,
                    "full_topk_from_scores: bad argument");
    DAISY_CHECK_ARG(workspace_bytes >= daisy_mf_full_rank_workspace_bytes(item_num),
                    "full_topk_from_scores: workspace too small");
    hipStream_t s = S(stream);
    const size_t n = (size_t)item_num;
    char *w = (char *)workspace;
    w += align_up(n * 4);
    float *sorted_scores = (float *)w;   w += align_up(n * 4);
    int64_t *ids = (int64_t *)w;         w += align_up(n * 8);
    int64_t *sorted_ids = (int64_t *)w;  w += align_up(n * 8);
    hipLaunchKernelGGL(k_iota_i64, dim3(grid_for(item_num, kBlock)), dim3(kBlock), 0, s, item_num, ids);
    DAISY_LAUNCH_CHECK();
    int rc = sort_pairs_desc_f32_i64(w, sort_pairs_desc_f32_i64_temp_bytes(n), scores, sorted_scores, ids,
                                     sorted_ids, item_num, s);
    if (rc) return rc;
    DAISY_HIP(hipMemcpyAsync(out_ids, sorted_ids, (size_t)topk * 8, hipMemcpyDeviceToDevice, s));
    return DAISY_OK;
}

}  // extern "C"
