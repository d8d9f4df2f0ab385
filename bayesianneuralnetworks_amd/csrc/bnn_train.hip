// bnn_train.hip -- the two callers either side of the backward path in the reference's training
// loop (examples/MNIST/train.py:53-65) that are otherwise a swarm of tiny launches:
//
//   * torch.optim.Adam(model.parameters()).step()  (train.py:41,65): ONE launch over all parameter
//     tensors (the capturable torch path is ~40 launches per step for the 12 tensors of the MLP);
//   * torch.nn.CrossEntropyLoss()(pred, y) over the (S * B, C) logits (train.py:39,59-61): loss and
//     d loss / d logits in one pass, one wave per 64 rows.
#include "bnn_device.hpp"

namespace bnn {

constexpr int kAdamThreads = 256;
constexpr int kAdamChunk = kAdamThreads * 8;          // scalars per workgroup
constexpr int kAdamMaxPerLaunch = 48;

struct AdamTensorDev {
    float *p;
    const float *g;
    float *m, *v;
    int64_t n;
    int32_t first_block;
    int32_t pad;
};
struct AdamLaunch {
    AdamTensorDev t[kAdamMaxPerLaunch];
    int32_t ntensors;
    float lr, beta1, beta2, eps, weight_decay;
};

// torch.optim.Adam (amsgrad = False, maximize = False, L2 weight_decay):
//   g += wd * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t = *step + 1 (device counter, so a captured graph advances it on every replay); the counter is
// bumped by a 1-thread tail kernel AFTER the update so that every workgroup reads the same t.
__global__ __launch_bounds__(kAdamThreads) void k_adam(AdamLaunch L, const float *__restrict__ step)
{
    int t = 0;
    for (int i = 1; i < L.ntensors; ++i)
        if ((int)blockIdx.x >= L.t[i].first_block) t = i;
    const AdamTensorDev T = L.t[t];
    const float tt = step[0] + 1.0f;
    const float bc1 = 1.0f - powf(L.beta1, tt);
    const float bc2 = 1.0f - powf(L.beta2, tt);
    const float step_size = L.lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    const int64_t base = (int64_t)(blockIdx.x - T.first_block) * kAdamChunk;
    const bool vec = ((reinterpret_cast<uintptr_t>(T.p) | reinterpret_cast<uintptr_t>(T.g) |
                       reinterpret_cast<uintptr_t>(T.m) | reinterpret_cast<uintptr_t>(T.v)) & 15u) == 0;
    auto upd = [&](float &p, float g, float &m, float &v) {
        g = fmaf(L.weight_decay, p, g);
        m = fmaf(L.beta1, m, (1.0f - L.beta1) * g);
        v = fmaf(L.beta2, v, (1.0f - L.beta2) * g * g);
        const float denom = sqrtf(v) * inv_sqrt_bc2 + L.eps;
        p -= step_size * (m / denom);
    };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int64_t e = base + ((int64_t)it * kAdamThreads + threadIdx.x) * 4;
        if (vec && e + 4 <= T.n) {
            float4 p = *reinterpret_cast<float4 *>(T.p + e);
            const float4 g = *reinterpret_cast<const float4 *>(T.g + e);
            float4 m = *reinterpret_cast<float4 *>(T.m + e);
            float4 v = *reinterpret_cast<float4 *>(T.v + e);
            upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
            *reinterpret_cast<float4 *>(T.p + e) = p;
            *reinterpret_cast<float4 *>(T.m + e) = m;
            *reinterpret_cast<float4 *>(T.v + e) = v;
        } else {
            for (int j = 0; j < 4; ++j)
                if (e + j < T.n) upd(T.p[e + j], T.g[e + j], T.m[e + j], T.v[e + j]);
        }
    }
}

// (+ the device epoch of the eps generator, when the optimizer step is also the end of the training step: one 1-thread
// launch instead of two -- each costs ~4 us)
__global__ void k_adam_step_bump(float *step, uint32_t *epoch_dev, uint32_t inc)
{
    step[0] += 1.0f;
    if (epoch_dev) epoch_dev[0] += inc;
}

// Mean cross-entropy over R rows of C <= 64 logits each (CrossEntropyLoss, reduction = 'mean'):
//   loss = mean_r (logsumexp(x_r) - x_r[y_r]);   gx_r = (softmax(x_r) - onehot(y_r)) / R
// One thread per row (rows are short: C = 10 for the north-star head); per-workgroup partial sums in
// double, fixed-order final sum by a second 1-workgroup kernel (bitwise reproducible).
__global__ __launch_bounds__(256) void k_xent_rows(const float *__restrict__ x, const int64_t *__restrict__ y,
                                                   float *__restrict__ gx, double *__restrict__ partial,
                                                   int64_t R, int C, float inv_R)
{
    __shared__ double red[4];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double li = 0.0;
    if (r < R) {
        const float *xr = x + r * C;
        float mx = xr[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, xr[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(xr[c] - mx);
        const float lse = mx + __logf(se);
        const int yc = (int)y[r];
        li = (double)(lse - xr[yc]);
        if (gx) {
            const float inv = inv_R / se;
            for (int c = 0; c < C; ++c) gx[r * C + c] = __expf(xr[c] - mx) * inv - (c == yc ? inv_R : 0.f);
        }
    }
    li = wave_sum(li);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = li;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_xent_final(const double *__restrict__ partial, int n, float inv_R, float *__restrict__ loss)
{
    __shared__ double red[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * (double)inv_R);
}

}  // namespace bnn

using namespace bnn;

extern "C" {

int bnn_adam_step(const bnn_adam_tensor_t *tensors, int ntensors, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float *step, void *stream)
{
    return bnn_adam_step_advance(tensors, ntensors, lr, beta1, beta2, eps, weight_decay, step, nullptr, 0, stream);
}

int bnn_adam_step_advance(const bnn_adam_tensor_t *tensors, int ntensors, float lr, float beta1, float beta2, float eps,
                          float weight_decay, float *step, uint32_t *advance_epoch, uint32_t advance_inc, void *stream)
{
    const char *who = "bnn_adam_step";
    if (!tensors || !step) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (ntensors < 1) { set_error("%s: ntensors < 1", who); return BNN_E_SHAPE; }
    if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(weight_decay >= 0.f)) {
        set_error("%s: hyper-parameter out of range", who);
        return BNN_E_RANGE;
    }
    for (int t = 0; t < ntensors; ++t) {
        if (!tensors[t].p || !tensors[t].g || !tensors[t].m || !tensors[t].v) { set_error("%s: tensor %d NULL", who, t); return BNN_E_NULL; }
        if (tensors[t].n < 1) { set_error("%s: tensor %d empty", who, t); return BNN_E_SHAPE; }
    }
    hipStream_t st = (hipStream_t)stream;
    for (int g0 = 0; g0 < ntensors; g0 += kAdamMaxPerLaunch) {
        AdamLaunch L{};
        const int cnt = ntensors - g0 < kAdamMaxPerLaunch ? ntensors - g0 : kAdamMaxPerLaunch;
        L.ntensors = cnt; L.lr = lr; L.beta1 = beta1; L.beta2 = beta2; L.eps = eps; L.weight_decay = weight_decay;
        int64_t blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const bnn_adam_tensor_t &s = tensors[g0 + i];
            L.t[i].p = s.p; L.t[i].g = s.g; L.t[i].m = s.m; L.t[i].v = s.v; L.t[i].n = s.n;
            L.t[i].first_block = (int32_t)blocks;
            blocks += (s.n + kAdamChunk - 1) / kAdamChunk;
        }
        if (blocks > 0x7FFFFFFF) { set_error("%s: too many elements", who); return BNN_E_RANGE; }
        hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(kAdamThreads), 0, st, L, step);
        const int rc = check_launch(who);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_adam_step_bump, dim3(1), dim3(1), 0, st, step, advance_epoch, advance_inc);
    return check_launch(who);
}

int64_t bnn_xent_workspace_bytes(int64_t rows) { return 8 * ((rows + 255) / 256 + 1); }

int bnn_softmax_xent(const float *logits, const int64_t *target, int64_t rows, int classes, float *loss,
                     float *g_logits, void *workspace, void *stream)
{
    const char *who = "bnn_softmax_xent";
    if (!logits || !target || !loss || !workspace) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (rows < 1 || classes < 1 || classes > 4096 || rows > ((int64_t)1 << 31) * 200) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = (rows + 255) / 256;
    if (nb > 0x7FFFFFFF) { set_error("%s: too many rows", who); return BNN_E_RANGE; }
    const float inv_R = (float)(1.0 / (double)rows);
    hipLaunchKernelGGL(k_xent_rows, dim3((unsigned)nb), dim3(256), 0, st, logits, target, g_logits,
                       reinterpret_cast<double *>(workspace), rows, classes, inv_R);
    int rc = check_launch(who);
    if (rc) return rc;
    hipLaunchKernelGGL(k_xent_final, dim3(1), dim3(256), 0, st, reinterpret_cast<const double *>(workspace), (int)nb, inv_R, loss);
    return check_launch(who);
}

}  // extern "C"
