// PyTorch-ROCm custom-op surface of the C ABI (BASELINE.json north_star: "surfaced to Python through
// PyTorch-ROCm custom ops"; SURVEY.md section 8b: "registered as torch ops on the HIP dispatch key").
//
// A SHIM with no logic of its own: every op checks its tensors (device, dtype, contiguity), takes raw device
// pointers and the current HIP stream, and calls the extern "C" entry point of libdaisyrec_hip.so named in its
// comment.  Tensors are borrowed, tables are mutated in place on the current stream (SURVEY 8b "Ownership").
// Registered under the CUDA dispatch key, which IS the HIP key in a ROCm build of PyTorch; there is no CPU
// kernel, so calling an op with host tensors fails in the dispatcher ("no kernel for CPU backend").
//
//   torch.ops.daisyrec.mf_predict(P, Q, u, i)                               daisy_mf_predict       MFRecommender.py:63-68
//   torch.ops.daisyrec.mf_rank_topk(P, Q, us, cands, topk)                  daisy_mf_rank_topk     MFRecommender.py:106-123
//   torch.ops.daisyrec.mf_full_rank(P, Q, u, topk)                          daisy_mf_full_rank     MFRecommender.py:126-133
//   torch.ops.daisyrec.sample_uniform_neg(indptr, items, I, num_ng, seed, epoch)
//                                                                           daisy_sample_neg_per_user   sampler.py:82-89
//   torch.ops.daisyrec.bpr_mf_step(P, Q, u, i, j, lr, reg_1, reg_2, gamma, loss_type)
//                                                                           daisy_bpr_set_batch + daisy_bpr_sgd_step
//                                                                           AbstractRecommender.py:119-128
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../include/daisyrec_amd.h"

namespace {

void need(const at::Tensor &t, at::ScalarType dt, const char *name) {
    TORCH_CHECK(t.is_cuda(), name, ": expected a HIP device tensor (there is no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == dt, name, ": expected dtype ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": tensor must be contiguous");
}

// every tensor of an op lives on the device of its first one; that device becomes the current one for the call, so
// the library's allocations (contexts), at::empty workspaces and kernel launches all land there
void same_device(const at::Tensor &first, std::initializer_list<const at::Tensor *> rest, const char *op) {
    for (const at::Tensor *t : rest)
        TORCH_CHECK(t->device() == first.device(), op, ": all tensors must be on one device (", first.device(), " vs ",
                    t->device(), ")");
}

void ok(int rc) { TORCH_CHECK(rc == DAISY_OK, "daisyrec: ", daisy_last_error()); }

daisy_stream_t stream_of(const at::Tensor &t) {
    return reinterpret_cast<daisy_stream_t>(c10::hip::getCurrentHIPStream(t.get_device()).stream());
}

at::Tensor mf_predict(const at::Tensor &P, const at::Tensor &Q, const at::Tensor &u, const at::Tensor &i) {
    need(P, at::kFloat, "P"); need(Q, at::kFloat, "Q"); need(u, at::kLong, "u"); need(i, at::kLong, "i");
    TORCH_CHECK(P.dim() == 2 && Q.dim() == 2 && P.size(1) == Q.size(1) && u.numel() == i.numel(), "mf_predict: shapes");
    same_device(P, {&Q, &u, &i}, "mf_predict");
    const c10::OptionalDeviceGuard guard(P.device());
    at::Tensor out = at::empty({u.numel()}, P.options());
    if (u.numel() == 0) return out;
    ok(daisy_mf_predict(P.data_ptr<float>(), Q.data_ptr<float>(), (int32_t)P.size(1), u.data_ptr<int64_t>(),
                        i.data_ptr<int64_t>(), u.numel(), out.data_ptr<float>(), stream_of(P)));
    return out;
}

at::Tensor mf_rank_topk(const at::Tensor &P, const at::Tensor &Q, const at::Tensor &us, const at::Tensor &cands,
                        int64_t topk) {
    need(P, at::kFloat, "P"); need(Q, at::kFloat, "Q"); need(us, at::kLong, "us"); need(cands, at::kLong, "cands");
    TORCH_CHECK(cands.dim() == 2 && us.numel() == cands.size(0), "mf_rank_topk: cands must be [len(us), C]");
    same_device(P, {&Q, &us, &cands}, "mf_rank_topk");
    const c10::OptionalDeviceGuard guard(P.device());
    const int64_t B = cands.size(0), C = cands.size(1);
    topk = topk < C ? topk : C;                                  // rank_list[:, :topk] truncates
    at::Tensor out = at::empty({B, topk}, cands.options());
    const size_t wb = daisy_mf_rank_workspace_bytes(B, C);
    at::Tensor ws = at::empty({(int64_t)(wb > 256 ? wb : 256)}, P.options().dtype(at::kByte));
    ok(daisy_mf_rank_topk(P.data_ptr<float>(), Q.data_ptr<float>(), (int32_t)P.size(1), us.data_ptr<int64_t>(),
                          cands.data_ptr<int64_t>(), B, C, (int32_t)topk, out.data_ptr<int64_t>(), nullptr,
                          ws.data_ptr(), (size_t)ws.numel(), stream_of(P)));
    return out;
}

at::Tensor mf_full_rank(const at::Tensor &P, const at::Tensor &Q, int64_t u, int64_t topk) {
    need(P, at::kFloat, "P"); need(Q, at::kFloat, "Q");
    same_device(P, {&Q}, "mf_full_rank");
    const c10::OptionalDeviceGuard guard(P.device());
    const int64_t I = Q.size(0);
    topk = topk < I ? topk : I;
    at::Tensor out = at::empty({topk}, P.options().dtype(at::kLong));
    const size_t wb = daisy_mf_full_rank_workspace_bytes(I);
    at::Tensor ws = at::empty({(int64_t)(wb > 256 ? wb : 256)}, P.options().dtype(at::kByte));
    ok(daisy_mf_full_rank(P.data_ptr<float>(), Q.data_ptr<float>(), (int32_t)P.size(1), I, u, (int32_t)topk,
                          out.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(), stream_of(P)));
    return out;
}

at::Tensor sample_uniform_neg(const at::Tensor &indptr, const at::Tensor &items, int64_t item_num, int64_t num_ng,
                              int64_t seed, int64_t epoch) {
    need(indptr, at::kLong, "indptr"); need(items, at::kInt, "items");
    same_device(items, {&indptr}, "sample_uniform_neg");
    const c10::OptionalDeviceGuard guard(items.device());
    const int64_t U = indptr.numel() - 1;
    at::Tensor js = at::empty({U, num_ng}, items.options());
    ok(daisy_sample_neg_per_user(indptr.data_ptr<int64_t>(), items.data_ptr<int32_t>(), U, item_num, (int32_t)num_ng,
                                 (uint64_t)seed, (uint64_t)epoch, js.data_ptr<int32_t>(), stream_of(items)));
    return js;
}

// per-(device, shape) training contexts of bpr_mf_step: the scratch a step needs lives in a daisy_bpr_ctx
struct CtxEntry {
    daisy_bpr_ctx *ctx;
    at::Tensor gQ, stats;
};
std::mutex g_mu;
std::map<std::tuple<int, int64_t, int64_t, int64_t, int64_t>, CtxEntry> g_ctx;

at::Tensor bpr_mf_step(at::Tensor P, at::Tensor Q, const at::Tensor &u, const at::Tensor &i, const at::Tensor &j,
                       double lr, double reg_1, double reg_2, double gamma, int64_t loss_type) {
    need(P, at::kFloat, "P"); need(Q, at::kFloat, "Q");
    need(u, at::kInt, "u"); need(i, at::kInt, "i"); need(j, at::kInt, "j");
    TORCH_CHECK(P.dim() == 2 && Q.dim() == 2 && P.size(1) == Q.size(1), "bpr_mf_step: tables must be [rows, d]");
    const int64_t B = u.numel();
    TORCH_CHECK(B > 0 && i.numel() == B && j.numel() == B, "bpr_mf_step: u, i, j must have the same length");
    same_device(P, {&Q, &u, &i, &j}, "bpr_mf_step");
    const c10::OptionalDeviceGuard guard(P.device());
    int64_t cap = 256;
    while (cap < B) cap *= 2;
    CtxEntry *e;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto key = std::make_tuple((int)P.get_device(), cap, P.size(1), P.size(0), Q.size(0));
        auto it = g_ctx.find(key);
        if (it == g_ctx.end()) {
            CtxEntry n;
            ok(daisy_bpr_ctx_create(&n.ctx, cap, (int32_t)P.size(1), P.size(0), Q.size(0)));
            n.gQ = at::zeros_like(Q);
            n.stats = at::zeros({DAISY_STATS_LEN}, P.options().dtype(at::kDouble));
            it = g_ctx.emplace(key, n).first;
        }
        e = &it->second;
    }
    daisy_stream_t s = stream_of(P);
    ok(daisy_bpr_ctx_set_pointwise(e->ctx, loss_type >= DAISY_LOSS_CL));
    ok(daisy_bpr_set_batch(e->ctx, u.data_ptr<int32_t>(), i.data_ptr<int32_t>(), j.data_ptr<int32_t>(), B, s));
    ok(daisy_bpr_ctx_validate_batch(e->ctx, s));                 // the reference raises IndexError here
    ok(daisy_bpr_ctx_invalidate_cache(e->ctx));                  // P may have been changed by other ops between calls
    at::Tensor loss = at::empty({}, P.options().dtype(at::kDouble));
    ok(daisy_bpr_sgd_step(e->ctx, P.data_ptr<float>(), Q.data_ptr<float>(), (int32_t)loss_type, (float)gamma, (float)lr,
                          (float)reg_1, (float)reg_2, e->gQ.data_ptr<float>(), e->stats.data_ptr<double>(), nullptr,
                          loss.data_ptr<double>(), DAISY_ITEM_FUSED, s));
    return loss;
}

}  // namespace

TORCH_LIBRARY(daisyrec, m) {
    m.def("mf_predict(Tensor P, Tensor Q, Tensor u, Tensor i) -> Tensor");
    m.def("mf_rank_topk(Tensor P, Tensor Q, Tensor us, Tensor cands, int topk) -> Tensor");
    m.def("mf_full_rank(Tensor P, Tensor Q, int u, int topk) -> Tensor");
    m.def("sample_uniform_neg(Tensor indptr, Tensor items, int item_num, int num_ng, int seed, int epoch) -> Tensor");
    m.def("bpr_mf_step(Tensor(a!) P, Tensor(b!) Q, Tensor u, Tensor i, Tensor j, float lr, float reg_1, float reg_2, "
          "float gamma, int loss_type) -> Tensor");
}

TORCH_LIBRARY_IMPL(daisyrec, CUDA, m) {      // the CUDA dispatch key is the HIP key of a ROCm build
    m.impl("mf_predict", &mf_predict);
    m.impl("mf_rank_topk", &mf_rank_topk);
    m.impl("mf_full_rank", &mf_full_rank);
    m.impl("sample_uniform_neg", &sample_uniform_neg);
    m.impl("bpr_mf_step", &bpr_mf_step);
}
