/*
 * bnn_oracle.c -- CPU restatement of the reference's variational-layer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bayesianneuralnetworks_amd/ may
 * import, link or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below that restates reference arithmetic against the .npz fixtures under tests/golden/, which
 * tests/golden/make_golden.py produced by importing the real reference
 * (pytorch_bayesian 0.0.4 from /root/reference) in the build container.
 *
 * Reference citations are relative to /root/reference/.
 *
 * Two kinds of function live here:
 *   (R) restatements of reference arithmetic (orc_sigma, orc_sample_affine,
 *       orc_linear, orc_conv2d, orc_kl_*, backward formulas);
 *   (B) the CPU twin of the build's own counter-based eps generator
 *       (orc_philox4x32_10, orc_eps4, orc_eps_fill).  The reference draws eps
 *       from torch's global mt19937 (pytorch_bayesian/nn/core.py:45), which a
 *       GPU kernel cannot reproduce; (B) pins the production Philox mode of the
 *       HIP kernels instead, while (R) with external eps pins them to the
 *       reference.
 *
 * Accumulations run in double and round once to float so that the oracle is
 * the more accurate side of every comparison.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------ (R) -- */

/* WeightNormal.stddev, pytorch_bayesian/nn/core.py:25-27:
 *   1e-10 + softplus(scale); torch softplus is beta=1, threshold=20
 *   (x > 20 -> x, else log1p(exp(x))), evaluated in fp32. */
float orc_sigma(float rho)
{
    float sp = (rho > 20.0f) ? rho : log1pf(expf(rho));
    return 1e-10f + sp;
}

void orc_sigma_vec(const float *rho, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_sigma(rho[i]);
}

/* WeightNormal.sample, pytorch_bayesian/nn/core.py:44-45:
 *   sampled = mean + stddev * eps   (eps handed in by the caller). */
void orc_sample_affine(const float *mu, const float *rho, const float *eps,
                       int64_t n, float *w)
{
    for (int64_t i = 0; i < n; ++i) {
        float s = orc_sigma(rho[i]);
        w[i] = mu[i] + s * eps[i];
    }
}

/* NormalLinear.forward, pytorch_bayesian/nn/dense.py:56-60:
 *   y = F.linear(x, w, b): x (M,K) row-major, w (N,K) row-major, b (N) or NULL. */
void orc_linear(const float *x, const float *w, const float *b,
                int64_t M, int64_t N, int64_t K, float *y)
{
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            double acc = b ? (double)b[n] : 0.0;
            const float *xr = x + m * K, *wr = w + n * K;
            for (int64_t k = 0; k < K; ++k) acc += (double)xr[k] * (double)wr[k];
            y[m * N + n] = (float)acc;
        }
}

/* NormalConv2d.forward, pytorch_bayesian/nn/conv.py:112-119:
 *   F.conv2d(x, w, b, stride, padding, dilation, groups), NCHW / OIHW. */
void orc_conv2d(const float *x, const float *w, const float *b,
                int64_t B, int64_t C, int64_t H, int64_t W,
                int64_t O, int64_t KH, int64_t KW,
                int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                int64_t dh, int64_t dw, int64_t groups, float *y)
{
    int64_t OH = (H + 2 * ph - dh * (KH - 1) - 1) / sh + 1;
    int64_t OW = (W + 2 * pw - dw * (KW - 1) - 1) / sw + 1;
    int64_t Cg = C / groups, Og = O / groups;
    for (int64_t n = 0; n < B; ++n)
        for (int64_t o = 0; o < O; ++o) {
            int64_t g = o / Og;
            for (int64_t oh = 0; oh < OH; ++oh)
                for (int64_t ow = 0; ow < OW; ++ow) {
                    double acc = b ? (double)b[o] : 0.0;
                    for (int64_t c = 0; c < Cg; ++c)
                        for (int64_t kh = 0; kh < KH; ++kh) {
                            int64_t ih = oh * sh - ph + kh * dh;
                            if (ih < 0 || ih >= H) continue;
                            for (int64_t kw = 0; kw < KW; ++kw) {
                                int64_t iw = ow * sw - pw + kw * dw;
                                if (iw < 0 || iw >= W) continue;
                                acc += (double)x[((n * C + g * Cg + c) * H + ih) * W + iw] *
                                       (double)w[((o * Cg + c) * KH + kh) * KW + kw];
                            }
                        }
                    y[((n * O + o) * OH + oh) * OW + ow] = (float)acc;
                }
        }
}

/* KLDivergence.compute_kl, pytorch_bayesian/nn/loss.py:16-28, with
 * torch.distributions.kl._kl_normal_normal:
 *   var_ratio = (sigma/sigma_p)^2 ; t1 = ((mu - mu_p)/sigma_p)^2
 *   kl = 0.5 * (var_ratio + t1 - 1 - log(var_ratio))
 * Returns the SUM over the tensor (the reference's .mean() = sum / n). */
double orc_kl_sum(const float *mu, const float *rho, int64_t n,
                  float prior_mu, float prior_sigma)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float sg = orc_sigma(rho[i]);
        float r0 = sg / prior_sigma;
        float vr = r0 * r0;
        float t0 = (mu[i] - prior_mu) / prior_sigma;
        float t1 = t0 * t0;
        float kl = 0.5f * (vr + t1 - 1.0f - logf(vr));
        s += (double)kl;
    }
    return s;
}

/* KLDivergence.forward, pytorch_bayesian/nn/loss.py:30-38:
 *   stack([kl_t.mean() for every weight/bias tensor]).mean() / n_batches. */
float orc_kl_divergence(const double *sums, const int64_t *numels, int64_t ntensors,
                        double n_batches)
{
    double acc = 0.0;
    for (int64_t t = 0; t < ntensors; ++t)
        acc += (double)(float)(sums[t] / (double)numels[t]);
    return (float)((acc / (double)ntensors) / n_batches);
}

/* Autograd of core.py:44-45 (what loss.backward() yields in the reference):
 *   dL/dmu = dL/dw ; dL/drho = dL/dw * eps * sigmoid(rho)
 *   (d softplus / d rho = sigmoid(rho); 1 above the threshold 20). */
void orc_sample_affine_bwd(const float *gw, const float *rho, const float *eps,
                           int64_t n, float *gmu, float *grho)
{
    for (int64_t i = 0; i < n; ++i) {
        float sig = (rho[i] > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-rho[i]));
        gmu[i] = gw[i];
        grho[i] = gw[i] * eps[i] * sig;
    }
}

/* Autograd of loss.py:28 for one tensor, scaled by `scale`
 * (= 1 / (numel * ntensors * n_batches) for loss.py:28,38):
 *   dkl/dmu  = (mu - mu_p) / sigma_p^2
 *   dkl/drho = (sigma/sigma_p^2 - 1/sigma) * sigmoid(rho) */
void orc_kl_bwd(const float *mu, const float *rho, int64_t n,
                float prior_mu, float prior_sigma, float scale,
                float *gmu, float *grho)
{
    double ps2 = (double)prior_sigma * (double)prior_sigma;
    for (int64_t i = 0; i < n; ++i) {
        double sg = (double)orc_sigma(rho[i]);
        double sig = (rho[i] > 20.0f) ? 1.0 : 1.0 / (1.0 + exp(-(double)rho[i]));
        gmu[i] = (float)(scale * ((double)mu[i] - prior_mu) / ps2);
        grho[i] = (float)(scale * (sg / ps2 - 1.0 / sg) * sig);
    }
}

/* ------------------------------------------------------------------ (B) -- */

/* Philox4x32-10 (Salmon et al., SC'11), the published algorithm. */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* The build's eps stream (include/bnn_hip.h, "RNG contract"):
 *   key     = (seed_lo, seed_hi)
 *   counter = (block, (stream << 16) | sample, epoch_host, epoch_dev)
 *   block   = element_index / 4, the four outputs feed elements 4*block..+3
 *   u       = ((x >> 8) + 0.5) * 2^-24   (fp32, round-to-nearest-even)
 *   Box-Muller: (x0,x1) -> z0 = r cos t, z1 = r sin t ; (x2,x3) -> z2, z3
 *               r = sqrt(-2 ln u_a), t = 2 pi u_b.
 * Evaluated here in double and rounded once. */
static inline float u01(uint32_t x)
{
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

void orc_eps4(uint64_t seed, uint32_t block, uint32_t stream, uint32_t sample,
              uint32_t epoch_host, uint32_t epoch_dev, float z[4])
{
    uint32_t ctr[4] = { block, (stream << 16) | (sample & 0xFFFFu), epoch_host, epoch_dev };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t x[4];
    orc_philox4x32_10(ctr, key, x);
    for (int p = 0; p < 2; ++p) {
        double ua = (double)u01(x[2 * p]);
        double ub = (double)u01(x[2 * p + 1]);
        double r = sqrt(-2.0 * log(ua));
        double t = 6.283185307179586476925286766559 * ub;
        z[2 * p] = (float)(r * cos(t));
        z[2 * p + 1] = (float)(r * sin(t));
    }
}

/* R rounds of the same function: R = 7 is Philox4x32-7 (known answers in Random123's kat_vectors). */
void orc_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* BNN_GEN_PHILOX7_U16 (include/bnn_hip.h, "RNG contract"): EIGHT eps per Philox4x32-7 block,
 *   counter = (element_index / 8, (stream << 16) | sample, epoch_host, epoch_dev),
 *   word x_k -> elements 2 k, 2 k + 1 of the block: ua = ((x_k & 0xffff) + 0.5) 2^-16, ub = ((x_k >> 16) + 0.5) 2^-16,
 *   z_even = r cos t, z_odd = r sin t, r = sqrt(-2 ln ua), t = 2 pi ub.  Evaluated in double, rounded once. */
void orc_eps8_u16(uint64_t seed, uint32_t block8, uint32_t stream, uint32_t sample,
                  uint32_t epoch_host, uint32_t epoch_dev, float z[8])
{
    uint32_t ctr[4] = { block8, (stream << 16) | (sample & 0xFFFFu), epoch_host, epoch_dev };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t x[4];
    orc_philox4x32_r(ctr, key, 7, x);
    for (int k = 0; k < 4; ++k) {
        double ua = ((double)(x[k] & 0xFFFFu) + 0.5) / 65536.0;
        double ub = ((double)(x[k] >> 16) + 0.5) / 65536.0;
        double r = sqrt(-2.0 * log(ua));
        double t = 6.283185307179586476925286766559 * ub;
        z[2 * k] = (float)(r * cos(t));
        z[2 * k + 1] = (float)(r * sin(t));
    }
}

/* eps of the stream `gen` names: 0 = BNN_GEN_PHILOX10_U24 (orc_eps4), 1 = BNN_GEN_PHILOX7_U16 (orc_eps8_u16). */
void orc_eps_fill_gen(uint64_t seed, uint32_t stream, uint32_t sample,
                      uint32_t epoch_host, uint32_t epoch_dev, int gen, int64_t n, float *eps)
{
    if (gen == 0) {
        for (int64_t b = 0; b * 4 < n; ++b) {
            float z[4];
            orc_eps4(seed, (uint32_t)b, stream, sample, epoch_host, epoch_dev, z);
            for (int j = 0; j < 4 && b * 4 + j < n; ++j) eps[b * 4 + j] = z[j];
        }
        return;
    }
    for (int64_t b = 0; b * 8 < n; ++b) {
        float z[8];
        orc_eps8_u16(seed, (uint32_t)b, stream, sample, epoch_host, epoch_dev, z);
        for (int j = 0; j < 8 && b * 8 + j < n; ++j) eps[b * 8 + j] = z[j];
    }
}

void orc_eps_fill(uint64_t seed, uint32_t stream, uint32_t sample,
                  uint32_t epoch_host, uint32_t epoch_dev, int64_t n, float *eps)
{
    for (int64_t b = 0; b * 4 < n; ++b) {
        float z[4];
        orc_eps4(seed, (uint32_t)b, stream, sample, epoch_host, epoch_dev, z);
        for (int j = 0; j < 4 && b * 4 + j < n; ++j) eps[b * 4 + j] = z[j];
    }
}

/* bf16 round-to-nearest-even of an fp32 value (NaN kept NaN), returned as fp32.
 * Used to state the tolerance of the bf16 configuration against fp32. */
float orc_bf16_round(float v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) {
        u |= 0x00400000u;
        u &= 0xFFFF0000u;
    } else {
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    }
    memcpy(&v, &u, 4);
    return v;
}
